// rx_prog.hip -- launch programs: the forward / backward launch lists of a plan, recorded once and replayed from C.
//
// Why: a cfg2 train step is ~700 entry-point calls.  Issued one by one from the host language (ctypes) they cost ~8 ms of
// host time per step -- the 64^3 configuration (6.6 ms of GPU work) was host-bound, and HIP graphs are no way out on this
// two-stream schedule (a captured graph with ~120 cross-stream edges replays SLOWER than eager launches).
// A program is not a graph: replay issues the very same hipLaunchKernelGGL / hipEventRecord / hipStreamWaitEvent calls on the
// very same streams, only from a C loop over std::function objects that hold the arguments by value.
//
// Recording is "record AND execute": between rx_prog_begin and rx_prog_end every stream-taking entry point called on this
// thread runs as usual and appends itself (RX_RECORD in its body), so the recording pass is an ordinary step.  Streams are
// recorded by INDEX into the table given to rx_prog_begin and substituted from the table given to rx_prog_run.
// Cross-stream ordering inside a program uses the numbered events below (rx_event_record / rx_stream_wait), which are
// ordinary recordable entry points.  Nothing here allocates device memory or synchronises (rx_prog_run with `ms` != NULL,
// the profiling replay, is the one documented exception: it synchronises the streams to read its timers).
#define RX_NO_HOST_MACROS 1
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>

#include "rx_common.h"
#include "rx_prog.h"

struct RxCmd {
  RxCmdFn fn;
  int stream_idx;
  const char* name;       // entry point (__func__)
  const char* kernel;     // what rx_note_kernel reported while it ran at record time ("" if nothing)
};

struct rx_prog {
  std::vector<RxCmd> cmds;
  std::vector<void*> streams;       // recording-time stream table
  bool bad_stream = false;
  std::vector<hipEvent_t> t0, t1;   // timers of the profiling replay
};

static thread_local rx_prog* g_rec = nullptr;
static thread_local int g_depth = 0;
static thread_local int g_note_seq = 0;          // bumped by rx_note_kernel_seq (see rx_elementwise.hip)
static thread_local int g_scope_note0 = 0;
static thread_local size_t g_scope_cmd = (size_t)-1;

extern "C" const char* rx_last_conv_kernel(void);
int rx_note_seq(void);

RxRecScope::RxRecScope() {
  rec = g_rec != nullptr && g_depth == 0;
  if (g_depth == 0) {
    g_scope_note0 = rx_note_seq();
    g_scope_cmd = (size_t)-1;
  }
  ++g_depth;
}
RxRecScope::~RxRecScope() {
  --g_depth;
  if (g_depth == 0 && g_rec && g_scope_cmd != (size_t)-1 && g_scope_cmd < g_rec->cmds.size() && rx_note_seq() != g_scope_note0)
    g_rec->cmds[g_scope_cmd].kernel = rx_last_conv_kernel();      // string literals: the pointer stays valid
}

void rx_rec_push(RxCmdFn fn, void* stream, const char* name) {
  rx_prog* p = g_rec;
  if (!p) return;
  int idx = -1;
  for (size_t i = 0; i < p->streams.size(); ++i)
    if (p->streams[i] == stream) idx = (int)i;
  if (idx < 0) p->bad_stream = true;
  g_scope_cmd = p->cmds.size();
  p->cmds.push_back(RxCmd{std::move(fn), idx, name, ""});
}

extern "C" rx_prog* rx_prog_create(void) { return new rx_prog(); }
extern "C" void rx_prog_destroy(rx_prog* p) {
  if (!p) return;
  if (g_rec == p) g_rec = nullptr;
  for (hipEvent_t e : p->t0) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->t1) (void)hipEventDestroy(e);
  delete p;
}
extern "C" int rx_prog_begin(rx_prog* p, void* const* streams, int n_streams) {
  if (!p || !streams || n_streams < 1) RX_FAIL(RX_EINVAL, "rx_prog_begin: bad arguments");
  if (g_rec) RX_FAIL(RX_EINVAL, "rx_prog_begin: another program is being recorded on this thread");
  p->cmds.clear();
  p->streams.assign(streams, streams + n_streams);
  p->bad_stream = false;
  g_rec = p;
  return RX_OK;
}
extern "C" int rx_prog_end(rx_prog* p) {
  if (!p || g_rec != p) RX_FAIL(RX_EINVAL, "rx_prog_end: this program is not being recorded");
  g_rec = nullptr;
  if (p->bad_stream) {
    p->cmds.clear();
    RX_FAIL(RX_EINVAL, "rx_prog_end: a recorded call used a stream that is not in the table given to rx_prog_begin");
  }
  return RX_OK;
}
extern "C" int rx_prog_len(const rx_prog* p) { return p ? (int)p->cmds.size() : -1; }
extern "C" const char* rx_prog_cmd_name(const rx_prog* p, int i) {
  return (p && i >= 0 && i < (int)p->cmds.size()) ? p->cmds[i].name : "";
}
extern "C" const char* rx_prog_cmd_kernel(const rx_prog* p, int i) {
  return (p && i >= 0 && i < (int)p->cmds.size()) ? p->cmds[i].kernel : "";
}
extern "C" int rx_prog_cmd_stream(const rx_prog* p, int i) { return (p && i >= 0 && i < (int)p->cmds.size()) ? p->cmds[i].stream_idx : -1; }

extern "C" int rx_prog_run(rx_prog* p, int first, int last, void* const* streams, int n_streams, float* ms) {
  if (!p || !streams || g_rec == p) RX_FAIL(RX_EINVAL, "rx_prog_run: bad arguments");
  const int n = (int)p->cmds.size();
  if (last < 0 || last > n) last = n;
  if (first < 0 || first > last) RX_FAIL(RX_EINVAL, "rx_prog_run: bad range [%d, %d) of %d", first, last, n);
  if (n_streams < (int)p->streams.size()) RX_FAIL(RX_EINVAL, "rx_prog_run: %d streams given, the program was recorded with %zu", n_streams, p->streams.size());
  if (ms) {
    while ((int)p->t0.size() < n) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) RX_FAIL(RX_ELAUNCH, "rx_prog_run: hipEventCreate failed");
      p->t0.push_back(a), p->t1.push_back(b);
    }
  }
  for (int i = first; i < last; ++i) {
    RxCmd& c = p->cmds[i];
    void* s = streams[c.stream_idx];
    if (ms) (void)hipEventRecord(p->t0[i], (hipStream_t)s);
    const int rc = c.fn(s);
    if (ms) (void)hipEventRecord(p->t1[i], (hipStream_t)s);
    if (rc != RX_OK) return rc;       // rx_last_error() already describes it
  }
  if (ms) {
    for (int k = 0; k < n_streams; ++k) (void)hipStreamSynchronize((hipStream_t)streams[k]);
    for (int i = first; i < last; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, p->t0[i], p->t1[i]);
      ms[i - first] = t;
    }
  }
  return RX_OK;
}

// ---- numbered events: cross-stream ordering that can be recorded -----------------------------------------------------
// ~60 slots per plan (events are created on first use).  A plan returns its slots with rx_event_free when it is destroyed (a
// long-lived process that rebuilds plans -- `.to()`, many input shapes in inference -- used to run out); the table is
// guarded by a mutex because ctypes releases the GIL and the backward pass runs on autograd's thread.
#define RX_MAX_EVENTS 65536
static hipEvent_t g_events[RX_MAX_EVENTS];
static bool g_event_made[RX_MAX_EVENTS];
static bool g_event_live[RX_MAX_EVENTS];
static int g_event_next = 0;
static std::vector<int> g_event_freelist;
static std::mutex g_event_mu;

extern "C" int rx_event_new(void) {
  std::lock_guard<std::mutex> lock(g_event_mu);
  int slot;
  if (!g_event_freelist.empty()) {
    slot = g_event_freelist.back();
    g_event_freelist.pop_back();
  } else {
    if (g_event_next >= RX_MAX_EVENTS) RX_FAIL(RX_EINVAL, "rx_event_new: out of event slots");
    slot = g_event_next++;
  }
  g_event_live[slot] = true;
  return slot;
}
// give a slot back (its hipEvent_t, if one was created, is kept and re-used by the next owner: a wait on a recycled event that
// was last recorded by the previous owner waits for work that is long complete or, at worst, for work of the same device).
// The caller must not use the number afterwards; recorded programs that mention it must be destroyed first.
extern "C" int rx_event_free(int slot) {
  std::lock_guard<std::mutex> lock(g_event_mu);
  if (slot < 0 || slot >= g_event_next || !g_event_live[slot]) RX_FAIL(RX_EINVAL, "rx_event_free: slot %d is not in use", slot);
  g_event_live[slot] = false;
  g_event_freelist.push_back(slot);
  return RX_OK;
}
extern "C" int rx_event_slots_in_use(void) {
  std::lock_guard<std::mutex> lock(g_event_mu);
  return g_event_next - (int)g_event_freelist.size();
}
static hipEvent_t* event_slot(int slot) {
  std::lock_guard<std::mutex> lock(g_event_mu);
  if (slot < 0 || slot >= g_event_next || !g_event_live[slot]) return nullptr;
  if (!g_event_made[slot]) {
    // These events only order streams of ONE device (producer kernel -> consumer kernel; every kernel still ends with its own
    // device-scope release), nobody synchronises the HOST on them: created without the system-scope fence (no cache write-back /
    // invalidate towards the host when the event fires; cfg2 step 18.01 -> 17.84 ms).  RX_EVENT_FLAGS (A/B knob): 0 plain, 1 without
    // the system-scope fence (default), 2 device-scope release.
    static const int mode = [] { const char* v = getenv("RX_EVENT_FLAGS"); return v ? atoi(v) : 1; }();
    unsigned flags = hipEventDisableTiming | (mode == 1 ? hipEventDisableSystemFence : mode == 2 ? hipEventReleaseToDevice : 0u);
    if (hipEventCreateWithFlags(&g_events[slot], flags) != hipSuccess) return nullptr;
    g_event_made[slot] = true;
  }
  return &g_events[slot];
}
extern "C" int rx_event_record(int slot, void* stream) {
  RX_RECORD(stream, [=](void* s) { return rx_event_record(slot, s); });
  hipEvent_t* e = event_slot(slot);
  if (!e) RX_FAIL(RX_EINVAL, "rx_event_record: bad slot %d", slot);
  if (hipEventRecord(*e, (hipStream_t)stream) != hipSuccess) RX_FAIL(RX_ELAUNCH, "rx_event_record: hipEventRecord failed");
  return RX_OK;
}
// `stream` waits for the work captured by the LAST rx_event_record(slot) issued before this call (HIP snapshots the event at
// the time of the wait, so a slot can be re-recorded right afterwards)
extern "C" int rx_stream_wait(int slot, void* stream) {
  RX_RECORD(stream, [=](void* s) { return rx_stream_wait(slot, s); });
  hipEvent_t* e = event_slot(slot);
  if (!e) RX_FAIL(RX_EINVAL, "rx_stream_wait: bad slot %d", slot);
  if (hipStreamWaitEvent((hipStream_t)stream, *e, 0) != hipSuccess) RX_FAIL(RX_ELAUNCH, "rx_stream_wait: hipStreamWaitEvent failed");
  return RX_OK;
}

// ---- per-call-site caches for what the entry points used to ask the runtime / environment on EVERY launch --------------
const char* rx_getenv_cached(const char* name) {
  static thread_local std::unordered_map<const char*, const char*> cache;      // keyed by the literal's address
  auto it = cache.find(name);
  if (it != cache.end()) return it->second;
  const char* v = getenv(name);
  cache.emplace(name, v);
  return v;
}
hipError_t rx_func_attr_once(const void* fn, hipFuncAttribute attr, int value) {
  // largest value already set per (device, kernel): the attribute belongs to the kernel's code object ON ONE DEVICE, so a
  // second device of the same process must get its own call
  static thread_local std::map<std::pair<int, const void*>, int> done;
  if (attr != hipFuncAttributeMaxDynamicSharedMemorySize) return hipFuncSetAttribute(fn, attr, value);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const auto key = std::make_pair(dev, fn);
  auto it = done.find(key);
  if (it != done.end() && it->second >= value) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, attr, value);
  if (e == hipSuccess) done[key] = value;
  return e;
}
