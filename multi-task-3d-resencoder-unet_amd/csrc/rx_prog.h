// rx_prog.h -- launch programs (internal): every stream-taking entry point of librxunet can append itself, arguments by value,
// to the program being recorded on this thread (RX_RECORD at the top of its body) while it executes as usual; rx_prog_run
// replays the recorded calls from C.  See include/rxunet.h "launch programs" and rx_prog.hip.
#pragma once
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/rxunet.h"

typedef std::function<int(void*)> RxCmdFn;      // argument: the stream to enqueue on at replay

struct RxRecScope {         // one per entry-point invocation: only the OUTERMOST entry point of a call chain records itself
  bool rec;
  RxRecScope();
  ~RxRecScope();
};
void rx_rec_push(RxCmdFn fn, void* stream, const char* name);

// by-value copies of pointer-passed descriptors / small arrays for the lambda captures
struct RxActV {
  rx_act a;
  bool has;
  RxActV(const rx_act* p) : has(p != nullptr) {
    if (p) a = *p; else memset(&a, 0, sizeof(a));
  }
  const rx_act* p() const { return has ? &a : nullptr; }
};
struct RxI3V {
  int32_t v[3];
  RxI3V(const int32_t* k) { v[0] = k[0], v[1] = k[1], v[2] = k[2]; }
};
struct RxSeV {
  rx_se_params s;
  bool has;
  RxSeV(const rx_se_params* p) : has(p != nullptr) {
    if (p) s = *p; else memset(&s, 0, sizeof(s));
  }
  const rx_se_params* p() const { return has ? &s : nullptr; }
};

#define RX_RECORD(stream, ...) \
  RxRecScope rx_scope__;       \
  if (rx_scope__.rec) rx_rec_push(RxCmdFn(__VA_ARGS__), (stream), __func__)
