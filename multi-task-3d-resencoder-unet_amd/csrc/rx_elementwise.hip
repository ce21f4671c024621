// rx_elementwise.hip -- HBM-bound kernels of the hot path: InstanceNorm statistics, the fused
// InstanceNorm-apply + LeakyReLU + residual-add forward / backward, AvgPool, per-channel sums,
// the 1x1x1 task head, the Cin<=4 stem convolution and the weight packers.
//
// All activations are channels-last; every thread moves 16-byte channel vectors (8 bf16 / 4 f32),
// adjacent lanes touch adjacent addresses, per-(n,c) reductions are deterministic two-stage
// reductions (per-block partials in a workspace, finalised in fp64) -- no float atomics.
#include <stdarg.h>

#include <math.h>

#include "rx_common.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void rx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* rx_last_error(void) { return g_err; }
static thread_local const char* g_last_kernel = "";
static thread_local int g_note_seq_ = 0;
void rx_note_kernel(const char* name) { g_last_kernel = name; ++g_note_seq_; }
int rx_note_seq(void) { return g_note_seq_; }
extern "C" const char* rx_last_conv_kernel(void) { return g_last_kernel; }
extern "C" int rx_abi_version(void) { return 1; }
extern "C" int rx_device_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// column (per-(n,c)) reductions: shared machinery
// ---------------------------------------------------------------------------------------------
// Work split: grid = (nchunks, N).  A block owns voxels [chunk*chunk_vox, ...) of sample n.
// thread -> (vl = tid / CV, cv = tid % CV): channel vector cv of voxels vl, vl+VP, ...
// Partials: partial[((n*nchunks + chunk)*NACC + a)*C + c].
struct ReducePlan {
  int nchunks, chunk_vox;
};
static inline ReducePlan rx_reduce_plan(long V, int C, int per16) {
  int CV = C / per16;
  int VP = 256 / CV;
  if (VP < 1) VP = 1;
  // ~1024 blocks per launch keep 256 CUs streaming; a block should own >= 8 passes of VP voxels
  long nch = V / ((long)VP * 8);
  if (nch < 1) nch = 1;
  if (nch > 512) nch = 512;
  long cvx = (V + nch - 1) / nch;
  cvx = (cvx + VP - 1) / VP * VP;
  nch = (V + cvx - 1) / cvx;
  ReducePlan p;
  p.nchunks = (int)nch;
  p.chunk_vox = (int)cvx;
  return p;
}
static inline size_t rx_reduce_ws_bytes(int N, long V, int C, int nacc) {
  // sized for the finest element type (per16 = 4 gives the most chunks)
  ReducePlan p = rx_reduce_plan(V, C, 4);
  ReducePlan q = rx_reduce_plan(V, C, 8);
  int nch = p.nchunks > q.nchunks ? p.nchunks : q.nchunks;
  return (size_t)N * nch * nacc * C * sizeof(float) + 256;
}

template <typename T, int NACC, typename Op>
__global__ __launch_bounds__(256) void colreduce_kernel(Op op, int V, int C, int chunk_vox, float* __restrict__ partial) {
  constexpr int P = Elem<T>::PER16;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [NACC][rows][C], rows = 4 (shuffle path) or VP
  const int CV = C / P;
  const int VP = 256 / CV > 0 ? 256 / CV : 1;
  const int tid = threadIdx.x;
  const int n = blockIdx.y, chunk = blockIdx.x;
  float acc[NACC][P];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int j = 0; j < P; ++j) acc[a][j] = 0.f;
  const int v_begin = chunk * chunk_vox;
  const int v_end = min(V, v_begin + chunk_vox);
  const int vl = tid / CV, cv = tid - vl * CV;
  if (vl < VP) {
    op.prepare(n, cv * P);
#pragma unroll 4
    for (int v = v_begin + vl; v < v_end; v += VP) op.accumulate(n, v, cv * P, acc);
  }
  const bool shuffle_path = (64 % CV) == 0;  // lanes of one wave with equal cv are CV apart
  int rows;
  if (shuffle_path) {
    for (int o = CV; o < 64; o <<= 1) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < P; ++j) acc[a][j] += __shfl_xor(acc[a][j], o, 64);
    }
    rows = 4;
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < CV) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < P; ++j) sm[(a * 4 + wave) * C + lane * P + j] = acc[a][j];
    }
  } else {
    rows = VP;
    if (vl < VP) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < P; ++j) sm[(a * VP + vl) * C + cv * P + j] = acc[a][j];
    }
  }
  __syncthreads();
  for (int i = tid; i < NACC * C; i += 256) {
    int a = i / C, c = i - a * C;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += sm[(a * rows + r) * C + c];
    partial[((size_t)(n * gridDim.x + chunk) * NACC + a) * C + c] = s;
  }
}

// finalize modes
enum { FIN_STATS = 0, FIN_MEAN2 = 1, FIN_SUM_OVER_N = 2 };
// FIN_STATS: out[n][c] = (mean, rstd) from (sum, sumsq);  FIN_MEAN2: out[n][c] = (s0/V, s1/V);
// FIN_SUM_OVER_N: out[a][c] = sum over n and chunks (NACC planes)
// one WORKGROUP per output element: the 256 threads stride over the chunks with up to four independent loads each in flight
// (a wave per element walked 512 chunks in 8 dependent round trips: 6.4 us per launch on average, 30 at worst, 71 launches per
// cfg2 step, every one of them between two kernels that depend on it), fp64 xor-shuffle combine, the four waves added in order.
__device__ inline void fin_gather(const float* __restrict__ base, size_t row_stride, int rows, int second, double& s0, double& s1) {
  const int tid = threadIdx.x;
  for (int k0 = tid; k0 < rows; k0 += 1024) {
    float v0[4], v1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 256 * u;
      const bool ok = k < rows;
      const float* p = base + (size_t)(ok ? k : 0) * row_stride;
      v0[u] = ok ? p[0] : 0.f;
      v1[u] = (ok && second) ? p[second] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s0 += (double)v0[u], s1 += (double)v1[u];
  }
  __shared__ double red[2][4];
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if ((tid & 63) == 0) red[0][tid >> 6] = s0, red[1][tid >> 6] = s1;
  __syncthreads();
  s0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  s1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
}
__global__ __launch_bounds__(256) void colreduce_finalize(const float* __restrict__ partial, int N, int nchunks, int nacc, int C, double V,
                                                          float eps, int mode, float* __restrict__ out) {
  const int i = blockIdx.x;  // element index
  double s0 = 0.0, s1 = 0.0;
  if (mode == FIN_SUM_OVER_N) {
    const int a = i / C, c = i - a * C;
    fin_gather(partial + (size_t)a * C + c, (size_t)nacc * C, N * nchunks, 0, s0, s1);
    if (threadIdx.x == 0) out[i] = (float)s0;
    return;
  }
  const int n = i / C, c = i - n * C;
  fin_gather(partial + ((size_t)n * nchunks * 2) * C + c, (size_t)2 * C, nchunks, C, s0, s1);
  if (threadIdx.x != 0) return;
  if (mode == FIN_STATS) {
    double mean = s0 / V;
    double var = s1 / V - mean * mean;
    if (var < 0.0) var = 0.0;
    out[2 * i] = (float)mean;
    out[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    out[2 * i] = (float)(s0 / V);
    out[2 * i + 1] = (float)(s1 / V);
  }
}

static inline void fin_launch(hipStream_t st, const float* partial, int N, int nchunks, int nacc, int C, double V, float eps, int mode,
                              float* out) {
  const int elems = mode == FIN_SUM_OVER_N ? nacc * C : N * C;
  hipLaunchKernelGGL(colreduce_finalize, dim3(elems), dim3(256), 0, st, partial, N, nchunks, nacc, C, V, eps, mode, out);
}

template <typename T>
struct ActView {
  const T* ptr;
  long sample_stride;  // elements
  int ld;
  __device__ inline const T* at(int n, int v, int c) const { return ptr + n * sample_stride + (long)v * ld + c; }
};
template <typename T>
static inline ActView<T> make_view(const rx_act* a) {
  ActView<T> r;
  r.ptr = (const T*)a->ptr;
  r.ld = a->ld;
  r.sample_stride = rx_act_voxels(a) * (long)a->ld;
  return r;
}

// ---- InstanceNorm statistics ----------------------------------------------------------------
template <typename T>
struct StatsOp {
  ActView<T> y;
  __device__ inline void prepare(int, int) {}
  __device__ inline void accumulate(int n, int v, int c0, float (&acc)[2][Elem<T>::PER16]) const {
    Vec16<T> x = ld16(y.at(n, v, c0));
#pragma unroll
    for (int j = 0; j < Elem<T>::PER16; ++j) {
      float f = Elem<T>::to_f(x.v[j]);
      acc[0][j] += f;
      acc[1][j] += f * f;
    }
  }
};

static int check_vec_channels(const rx_act* a, int dt, const char* who) {
  int per16 = dt == RX_F32 ? 4 : 8;
  if (!rx_act_ok(a)) RX_FAIL(RX_EINVAL, "%s: bad activation descriptor", who);
  if (a->c % per16 || a->ld % per16 || ((uintptr_t)a->ptr & 15)) RX_FAIL(RX_EUNSUPPORTED, "%s: channels/ld/ptr must be 16-byte multiples (c=%d ld=%d)", who, a->c, a->ld);
  if (a->c / per16 > 256) RX_FAIL(RX_EUNSUPPORTED, "%s: too many channels (%d)", who, a->c);
  return RX_OK;
}

extern "C" size_t rx_instnorm_stats_workspace(const rx_act* y) {
  if (!rx_act_ok(y)) return 0;
  return rx_reduce_ws_bytes(y->n, rx_act_voxels(y), y->c, 2);
}

extern "C" int rx_instnorm_stats(rx_dtype dt, const rx_act* y, float eps, float* stats, void* ws, size_t ws_bytes,
                                 void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y)](void* s) { return rx_instnorm_stats(dt, y_.p(), eps, stats, ws, ws_bytes, s); });
  int rc = check_vec_channels(y, dt, "rx_instnorm_stats");
  if (rc) return rc;
  if (!stats || !ws) RX_FAIL(RX_EINVAL, "rx_instnorm_stats: null stats/workspace");
  if (ws_bytes < rx_instnorm_stats_workspace(y)) RX_FAIL(RX_EWORKSPACE, "rx_instnorm_stats: workspace too small");
  const long V = rx_act_voxels(y);
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, y->c, P);
    int CV = y->c / P, VP = 256 / CV;
    StatsOp<T> op{make_view<T>(y)};
    size_t lds = (size_t)2 * (VP > 4 ? VP : 4) * y->c * sizeof(float);
    hipLaunchKernelGGL((colreduce_kernel<T, 2, StatsOp<T>>), dim3(p.nchunks, y->n), dim3(256), lds, st, op, (int)V, y->c,
                       p.chunk_vox, (float*)ws);
    int tot = y->n * y->c;
    fin_launch(st, (const float*)ws, y->n, p.nchunks, 2,
                       y->c, (double)V, eps, (int)FIN_STATS, stats);
  });
  RX_CHECK_LAUNCH("rx_instnorm_stats");
  return RX_OK;
}

// ---- nn.Dropout3d / nn.Dropout2d in front of an InstanceNorm (simple_conv_blocks.py:57-66: conv -> dropout -> norm) --------
// Channel dropout multiplies a whole (n, c) plane by 0 or by s = 1/(1-p).  InstanceNorm(affine=False) of s*y is
// (y - mean) / sqrt(var + eps/s^2): the kept planes need no pass over y at all, only the smaller eps (the caller passes
// eps*(1-p)^2 to the statistics); a dropped plane normalises to exactly 0, which is rstd = 0 in the (mean, rstd) pair every
// forward AND backward InstanceNorm kernel of this library works from (xhat = 0, dy = rstd * (...) = 0).  This entry point
// applies the second half: stats[i].rstd *= keep[i], keep[n*C + c] in {0, 1}.
__global__ __launch_bounds__(256) void stats_mask_kernel(float* __restrict__ stats, const float* __restrict__ keep, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) stats[2 * i + 1] *= keep[i];
}

extern "C" int rx_instnorm_stats_mask(float* stats, const float* keep, int count, void* stream) {
  RX_RECORD(stream, [=](void* s) { return rx_instnorm_stats_mask(stats, keep, count, s); });
  if (!stats || !keep || count < 0) RX_FAIL(RX_EINVAL, "rx_instnorm_stats_mask: bad arguments");
  if (count == 0) return RX_OK;
  hipLaunchKernelGGL(stats_mask_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, keep, count);
  RX_CHECK_LAUNCH("rx_instnorm_stats_mask");
  return RX_OK;
}

// (mean, rstd) from per-chunk partial sums laid out like colreduce_kernel's (rx_conv_halo.hip leaves such partials behind
// when a persistent conv kernel accumulates the statistics of its own output)
void rx_stats_finalize_launch(const float* partial, int N, int nchunks, int C, double V, float eps, float* stats, hipStream_t st) {
  fin_launch(st, partial, N, nchunks, 2, C, V, eps, (int)FIN_STATS, stats);
}

// ---- per-channel sum over (n, voxels) -------------------------------------------------------
template <typename T>
struct SumOp {
  ActView<T> x;
  __device__ inline void prepare(int, int) {}
  __device__ inline void accumulate(int n, int v, int c0, float (&acc)[1][Elem<T>::PER16]) const {
    Vec16<T> a = ld16(x.at(n, v, c0));
#pragma unroll
    for (int j = 0; j < Elem<T>::PER16; ++j) acc[0][j] += Elem<T>::to_f(a.v[j]);
  }
};
extern "C" size_t rx_channel_sum_workspace(const rx_act* x) {
  if (!rx_act_ok(x)) return 0;
  return rx_reduce_ws_bytes(x->n, rx_act_voxels(x), x->c, 1);
}
extern "C" int rx_channel_sum(rx_dtype dt, const rx_act* x, float* out, void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x)](void* s) { return rx_channel_sum(dt, x_.p(), out, ws, ws_bytes, s); });
  int rc = check_vec_channels(x, dt, "rx_channel_sum");
  if (rc) return rc;
  if (!out || !ws) RX_FAIL(RX_EINVAL, "rx_channel_sum: null out/workspace");
  if (ws_bytes < rx_channel_sum_workspace(x)) RX_FAIL(RX_EWORKSPACE, "rx_channel_sum: workspace too small");
  const long V = rx_act_voxels(x);
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, x->c, P);
    int CV = x->c / P, VP = 256 / CV;
    SumOp<T> op{make_view<T>(x)};
    size_t lds = (size_t)(VP > 4 ? VP : 4) * x->c * sizeof(float);
    hipLaunchKernelGGL((colreduce_kernel<T, 1, SumOp<T>>), dim3(p.nchunks, x->n), dim3(256), lds, st, op, (int)V, x->c,
                       p.chunk_vox, (float*)ws);
    fin_launch(st, (const float*)ws, x->n, p.nchunks, 1,
                       x->c, (double)V, 0.f, (int)FIN_SUM_OVER_N, out);
  });
  RX_CHECK_LAUNCH("rx_channel_sum");
  return RX_OK;
}

// ---- fused InstanceNorm-apply + residual + LeakyReLU forward -------------------------------
// grid = (G, N); a thread keeps its channel vector fixed (G*256 % CV == 0) so mean/rstd live in
// registers for the whole sweep.
template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void in_act_fwd_kernel(const T* __restrict__ y, int ldy, long sy, const float* __restrict__ stats,
                                                         const T* __restrict__ res, int ldr, long sr, T* __restrict__ out, int ldo,
                                                         long so, int V, int C, float slope) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)V * CV;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = stats[2 * ((size_t)n * C + cv * P + j)];
    rstd[j] = stats[2 * ((size_t)n * C + cv * P + j) + 1];
  }
  const T* yn = y + n * sy;
  const T* rn = HAS_RES ? res + n * sr : nullptr;
  T* on = out + n * so;
  for (; i < total; i += step) {
    long v = i / CV;
    Vec16<T> a = ld16(yn + v * ldy + cv * P);
    Vec16<T> r;
    if (HAS_RES) r = ld16(rn + v * ldr + cv * P);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = (Elem<T>::to_f(a.v[j]) - mean[j]) * rstd[j];
      if (HAS_RES) f += Elem<T>::to_f(r.v[j]);
      f = f > 0.f ? f : f * slope;
      o.v[j] = Elem<T>::from_f(f);
    }
    st16(on + v * ldo + cv * P, o);
  }
}

static inline int sweep_grid(long total_vec, int CV) {
  // number of blocks G with (G*256) % CV == 0, so that every thread keeps one channel vector
  int g = CV, d = 256;
  while (g % 2 == 0 && d > 1) {
    g /= 2;
    d /= 2;
  }
  long want = (total_vec + 256 * 8 - 1) / (256 * 8);
  if (want < 1) want = 1;
  if (want > 2048) want = 2048;
  long G = (want + g - 1) / g * g;
  return (int)G;
}

static int same_geom(const rx_act* a, const rx_act* b) {
  return a->n == b->n && a->z == b->z && a->y == b->y && a->x == b->x && a->c == b->c;
}

extern "C" int rx_instnorm_act_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* residual,
                                   const rx_act* out, float slope, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), residual_ = RxActV(residual), out_ = RxActV(out)](void* s) { return rx_instnorm_act_fwd(dt, y_.p(), stats, residual_.p(), out_.p(), slope, s); });
  int rc = check_vec_channels(y, dt, "rx_instnorm_act_fwd(y)");
  if (rc) return rc;
  rc = check_vec_channels(out, dt, "rx_instnorm_act_fwd(out)");
  if (rc) return rc;
  if (!stats || !same_geom(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_fwd: geometry mismatch / null stats");
  if (residual) {
    rc = check_vec_channels(residual, dt, "rx_instnorm_act_fwd(residual)");
    if (rc) return rc;
    if (!same_geom(y, residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_fwd: residual geometry mismatch");
  }
  const long V = rx_act_voxels(y);
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    int CV = y->c / P;
    int G = sweep_grid(V * CV, CV);
    if (residual)
      hipLaunchKernelGGL((in_act_fwd_kernel<T, true>), dim3(G, y->n), dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld, stats,
                         (const T*)residual->ptr, residual->ld, V * residual->ld, (T*)out->ptr, out->ld, V * out->ld, (int)V,
                         y->c, slope);
    else
      hipLaunchKernelGGL((in_act_fwd_kernel<T, false>), dim3(G, y->n), dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld, stats,
                         (const T*)nullptr, 0, 0L, (T*)out->ptr, out->ld, V * out->ld, (int)V, y->c, slope);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_fwd");
  return RX_OK;
}

// ---- fused backward --------------------------------------------------------------------------
// g' = g * (out > 0 ? 1 : slope);  xhat = (y-mean)*rstd
// pass 1: m1 = mean(g'), m2 = mean(g'*xhat) per (n,c);  pass 2: dy = rstd*(g' - m1 - xhat*m2)
template <typename T>
struct InBwdOp {
  ActView<T> g, y, out;
  const float* stats;
  int C;
  float slope;
  bool use_mask;   // LeakyReLU mask from the sign of the saved output (residual blocks)
  bool mask_xhat;  // no residual: out > 0 <=> xhat > 0, the output tensor is not read at all
  float mean[Elem<T>::PER16], rstd[Elem<T>::PER16];
  __device__ inline void prepare(int n, int c0) {
#pragma unroll
    for (int j = 0; j < Elem<T>::PER16; ++j) {
      mean[j] = stats[2 * ((size_t)n * C + c0 + j)];
      rstd[j] = stats[2 * ((size_t)n * C + c0 + j) + 1];
    }
  }
  __device__ inline void accumulate(int n, int v, int c0, float (&acc)[2][Elem<T>::PER16]) const {
    constexpr int P = Elem<T>::PER16;
    Vec16<T> gv = ld16(g.at(n, v, c0));
    Vec16<T> yv = ld16(y.at(n, v, c0));
    Vec16<T> ov;
    if (use_mask) ov = ld16(out.at(n, v, c0));
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(gv.v[j]);
      float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (use_mask && !(Elem<T>::to_f(ov.v[j]) > 0.f)) gg *= slope;
      if (mask_xhat && !(xh > 0.f)) gg *= slope;
      acc[0][j] += gg;
      acc[1][j] += gg * xh;
    }
  }
};

template <typename T, bool HAS_DRES, bool ACC_DRES>
__global__ __launch_bounds__(256) void in_act_bwd_apply_kernel(const T* __restrict__ g, int ldg, long sg, const T* __restrict__ y, int ldy,
                                                               long sy, const T* __restrict__ out, int ldo, long so,
                                                               const float* __restrict__ stats, const float* __restrict__ m12,
                                                               T* __restrict__ dy, int lddy, long sdy, T* __restrict__ dres, int lddr,
                                                               long sdr, int V, int C, float slope, int use_mask) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)V * CV;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P], m1[P], m2[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    size_t k = (size_t)n * C + cv * P + j;
    mean[j] = stats[2 * k];
    rstd[j] = stats[2 * k + 1];
    m1[j] = m12[2 * k];
    m2[j] = m12[2 * k + 1];
  }
  for (; i < total; i += step) {
    long v = i / CV;
    Vec16<T> gv = ld16(g + n * sg + v * ldg + cv * P);
    Vec16<T> yv = ld16(y + n * sy + v * ldy + cv * P);
    Vec16<T> ov;
    if (use_mask == 1) ov = ld16(out + n * so + v * ldo + cv * P);
    Vec16<T> dv, rv;
    if (HAS_DRES && ACC_DRES) rv = ld16(dres + n * sdr + v * lddr + cv * P);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(gv.v[j]);
      float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (use_mask == 1 && !(Elem<T>::to_f(ov.v[j]) > 0.f)) gg *= slope;
      if (use_mask == 2 && !(xh > 0.f)) gg *= slope;
      dv.v[j] = Elem<T>::from_f(rstd[j] * (gg - m1[j] - xh * m2[j]));
      if (HAS_DRES) {
        float r = gg;
        if (ACC_DRES) r += Elem<T>::to_f(rv.v[j]);
        rv.v[j] = Elem<T>::from_f(r);
      }
    }
    st16(dy + n * sdy + v * lddy + cv * P, dv);
    if (HAS_DRES) st16(dres + n * sdr + v * lddr + cv * P, rv);
  }
}

#define RX_LAUNCH_APPLY(HD, AD)                                                                                                   \
  hipLaunchKernelGGL((in_act_bwd_apply_kernel<T, HD, AD>), dim3(G, N), dim3(256), 0, st, (const T*)g->ptr, g->ld, V * g->ld,     \
                     (const T*)y->ptr, y->ld, V * y->ld, outp, ldo, V * ldo, stats, (const float*)m12, (T*)dy->ptr, dy->ld,       \
                     V * dy->ld, d_residual ? (T*)d_residual->ptr : (T*)nullptr, d_residual ? d_residual->ld : 0,                 \
                     d_residual ? V * d_residual->ld : 0L, (int)V, C, slope, use_mask ? 1 : (mask_xhat ? 2 : 0))

// ---- single-launch InstanceNorm forward / backward for SMALL tensors (low-resolution stages) ---------------------
// At 8^3 and below (measured: 16^3 is already better off with the chip-filling three-launch path; narrower 8-channel
// groups did not change that) a layer's tensor is a few hundred KB and the three launches (partials, finalize, apply) are pure
// launch latency on the critical chain.  One workgroup owns (sample n, 32 consecutive channels): pass 1 reduces over all
// voxels (thread = (voxel lane, 16-byte channel chunk); xor-shuffle across the 16 voxel lanes of a wave, LDS across the
// 4 waves, fp64 for the final combination), pass 2 re-reads the (L2-resident) data and applies.
template <typename T, int G, bool HAS_RES>
__global__ __launch_bounds__(256) void in_small_fwd_kernel(const T* __restrict__ y, int ldy, long sy, const T* __restrict__ res, int ldr,
                                                           long sr, T* __restrict__ out, int ldo, long so, float* __restrict__ stats, int V,
                                                           int C, float eps, float slope) {
  constexpr int P = Elem<T>::PER16;
  constexpr int CPG = G / P;  // 16-byte chunks per G-channel group
  __shared__ double red[4][2][G];
  __shared__ float mr[2][G];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, c0 = blockIdx.x * G;
  const int ck = tid % CPG, vl = tid / CPG;
  const int VL = 256 / CPG;
  const T* yn = y + n * sy + c0 + ck * P;
  float s[P], q[P];
#pragma unroll
  for (int j = 0; j < P; ++j) s[j] = q[j] = 0.f;
  for (int v = vl; v < V; v += VL) {
    Vec16<T> a = ld16(yn + (long)v * ldy);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = Elem<T>::to_f(a.v[j]);
      s[j] += f;
      q[j] += f * f;
    }
  }
  for (int o = CPG; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      s[j] += __shfl_xor(s[j], o, 64);
      q[j] += __shfl_xor(q[j], o, 64);
    }
  }
  if (lane < CPG) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      red[wave][0][lane * P + j] = (double)s[j];
      red[wave][1][lane * P + j] = (double)q[j];
    }
  }
  __syncthreads();
  if (tid < G) {
    double s0 = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
    double s1 = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
    double mean = s0 / V, var = s1 / V - mean * mean;
    if (var < 0.0) var = 0.0;
    float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    mr[0][tid] = m;
    mr[1][tid] = r;
    stats[2 * ((size_t)n * C + c0 + tid)] = m;
    stats[2 * ((size_t)n * C + c0 + tid) + 1] = r;
  }
  __syncthreads();
  float mean[P], rstd[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = mr[0][ck * P + j];
    rstd[j] = mr[1][ck * P + j];
  }
  const T* rn = HAS_RES ? res + n * sr + c0 + ck * P : nullptr;
  T* on = out + n * so + c0 + ck * P;
  for (int v = vl; v < V; v += VL) {
    Vec16<T> a = ld16(yn + (long)v * ldy);
    Vec16<T> r;
    if (HAS_RES) r = ld16(rn + (long)v * ldr);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = (Elem<T>::to_f(a.v[j]) - mean[j]) * rstd[j];
      if (HAS_RES) f += Elem<T>::to_f(r.v[j]);
      f = f > 0.f ? f : f * slope;
      o.v[j] = Elem<T>::from_f(f);
    }
    st16(on + (long)v * ldo, o);
  }
}

// mask_mode: 0 none, 1 sign of `out`, 2 sign of xhat
template <typename T, int G>
__global__ __launch_bounds__(256) void in_small_bwd_kernel(const T* __restrict__ g, int ldg, long sg, const T* __restrict__ y, int ldy, long sy,
                                                           const T* __restrict__ out, int ldo, long so, const float* __restrict__ stats,
                                                           T* __restrict__ dy, int lddy, long sdy, T* __restrict__ dres, int lddr, long sdr,
                                                           int acc_res, int V, int C, float slope, int mask_mode) {
  constexpr int P = Elem<T>::PER16;
  constexpr int CPG = G / P;
  __shared__ double red[4][2][G];
  __shared__ float mm[2][G];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, c0 = blockIdx.x * G;
  const int ck = tid % CPG, vl = tid / CPG;
  const int VL = 256 / CPG;
  const long co = c0 + ck * P;
  float mean[P], rstd[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = stats[2 * ((size_t)n * C + co + j)];
    rstd[j] = stats[2 * ((size_t)n * C + co + j) + 1];
  }
  auto gprime = [&](int v, float (&gg)[P], float (&xh)[P]) {
    Vec16<T> gv = ld16(g + n * sg + (long)v * ldg + co);
    Vec16<T> yv = ld16(y + n * sy + (long)v * ldy + co);
    Vec16<T> ov;
    if (mask_mode == 1) ov = ld16(out + n * so + (long)v * ldo + co);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      gg[j] = Elem<T>::to_f(gv.v[j]);
      xh[j] = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (mask_mode == 1 && !(Elem<T>::to_f(ov.v[j]) > 0.f)) gg[j] *= slope;
      if (mask_mode == 2 && !(xh[j] > 0.f)) gg[j] *= slope;
    }
  };
  float s[P], q[P];
#pragma unroll
  for (int j = 0; j < P; ++j) s[j] = q[j] = 0.f;
  for (int v = vl; v < V; v += VL) {
    float gg[P], xh[P];
    gprime(v, gg, xh);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      s[j] += gg[j];
      q[j] += gg[j] * xh[j];
    }
  }
  for (int o = CPG; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      s[j] += __shfl_xor(s[j], o, 64);
      q[j] += __shfl_xor(q[j], o, 64);
    }
  }
  if (lane < CPG) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      red[wave][0][lane * P + j] = (double)s[j];
      red[wave][1][lane * P + j] = (double)q[j];
    }
  }
  __syncthreads();
  if (tid < G) {
    mm[0][tid] = (float)((red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid]) / V);
    mm[1][tid] = (float)((red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid]) / V);
  }
  __syncthreads();
  float m1[P], m2[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    m1[j] = mm[0][ck * P + j];
    m2[j] = mm[1][ck * P + j];
  }
  for (int v = vl; v < V; v += VL) {
    float gg[P], xh[P];
    gprime(v, gg, xh);
    Vec16<T> dv, rv;
    if (dres && acc_res) rv = ld16(dres + n * sdr + (long)v * lddr + co);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      dv.v[j] = Elem<T>::from_f(rstd[j] * (gg[j] - m1[j] - xh[j] * m2[j]));
      if (dres) {
        float r = gg[j];
        if (acc_res) r += Elem<T>::to_f(rv.v[j]);
        rv.v[j] = Elem<T>::from_f(r);
      }
    }
    st16(dy + n * sdy + (long)v * lddy + co, dv);
    if (dres) st16(dres + n * sdr + (long)v * lddr + co, rv);
  }
}

// tuning knob (env RX_IN_SMALL_MAX, read once): largest per-sample voxel count that takes the single-launch path
static long rx_in_small_max() {
  static long v = [] { const char* e = getenv("RX_IN_SMALL_MAX"); return e ? atol(e) : 512L; }();
  return v;
}
#define RX_IN_SMALL_MAX_VOXELS rx_in_small_max()
// channels per workgroup of the single-launch kernels: 32, or 8 from RX_IN_SMALL_NARROW voxels per sample upwards (default 256:
// the 8^3 stage; 80 workgroups instead of 20 for 320 channels x 2 samples, one 16-byte vector per voxel and thread.  Alone
// 12.4 -> 7.6 us for the backward of a 320-channel 8^3 layer, 17.42 / 17.37 -> 17.32 / 17.30 ms per cfg2 step on one box)
static long rx_in_small_narrow() {
  static long v = [] { const char* e = getenv("RX_IN_SMALL_NARROW"); return e ? atol(e) : 256L; }();
  return v;
}

extern "C" int rx_instnorm_act_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out,
                                   float slope, const rx_act* dy, const rx_act* d_residual, int accumulate_residual, void* ws,
                                   size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, g_ = RxActV(g), y_ = RxActV(y), out_ = RxActV(out), dy_ = RxActV(dy), d_residual_ = RxActV(d_residual)](void* s) { return rx_instnorm_act_bwd(dt, g_.p(), y_.p(), stats, out_.p(), slope, dy_.p(), d_residual_.p(), accumulate_residual, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(g, dt, "rx_instnorm_act_bwd(g)"))) return rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_act_bwd(y)"))) return rc;
  if ((rc = check_vec_channels(dy, dt, "rx_instnorm_act_bwd(dy)"))) return rc;
  // mask source: none (slope 1) | sign of xhat (no residual: `out` may be NULL and is never read) | saved output
  const bool mask_xhat = slope != 1.0f && out == nullptr;
  const bool use_mask = slope != 1.0f && out != nullptr;
  if (use_mask) {
    if ((rc = check_vec_channels(out, dt, "rx_instnorm_act_bwd(out)"))) return rc;
    if (!same_geom(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd: out geometry mismatch");
  }
  if (d_residual) {
    if ((rc = check_vec_channels(d_residual, dt, "rx_instnorm_act_bwd(d_residual)"))) return rc;
    if (!same_geom(y, d_residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd: d_residual geometry mismatch");
  }
  if (!stats || !ws || !same_geom(y, g) || !same_geom(y, dy)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd: bad arguments");
  const long V = rx_act_voxels(y);
  const int N = y->n, C = y->c;
  if (V <= RX_IN_SMALL_MAX_VOXELS && C % 32 == 0) {   // low-resolution stages: one launch instead of three
    hipStream_t st1 = (hipStream_t)stream;
    const int mode = use_mask ? 1 : (mask_xhat ? 2 : 0);
    const bool narrow = dt != RX_F32 && V >= rx_in_small_narrow();   // (fp32 = parity mode: the summation order the goldens' seeds were screened with)
    dim3 grid1(narrow ? C / 8 : C / 32, N);
#define RX_LAUNCH_IN_SMALL_BWD(G)                                                                                                     \
  hipLaunchKernelGGL((in_small_bwd_kernel<T, G>), grid1, dim3(256), 0, st1, (const T*)g->ptr, g->ld, V * g->ld, (const T*)y->ptr, y->ld, \
                     V * y->ld, use_mask ? (const T*)out->ptr : (const T*)nullptr, use_mask ? out->ld : 0,                           \
                     use_mask ? V * out->ld : 0L, stats, (T*)dy->ptr, dy->ld, V * dy->ld,                                             \
                     d_residual ? (T*)d_residual->ptr : (T*)nullptr, d_residual ? d_residual->ld : 0,                                 \
                     d_residual ? V * d_residual->ld : 0L, accumulate_residual, (int)V, C, slope, mode)
    RX_DISPATCH_DTYPE(dt, T, {
      if (narrow) RX_LAUNCH_IN_SMALL_BWD(8);
      else RX_LAUNCH_IN_SMALL_BWD(32);
    });
    RX_CHECK_LAUNCH("rx_instnorm_act_bwd(small)");
    return RX_OK;
  }
  size_t need = rx_reduce_ws_bytes(N, V, C, 2) + (size_t)N * C * 2 * sizeof(float);
  if (ws_bytes < need) RX_FAIL(RX_EWORKSPACE, "rx_instnorm_act_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
  float* partial = (float*)ws;
  float* m12 = (float*)((char*)ws + rx_align_up(rx_reduce_ws_bytes(N, V, C, 2) - 256, 256));
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, C, P);
    int CV = C / P, VP = 256 / CV;
    InBwdOp<T> op{make_view<T>(g), make_view<T>(y), use_mask ? make_view<T>(out) : make_view<T>(y), stats, C, slope, use_mask, mask_xhat, {}, {}};
    size_t lds = (size_t)2 * (VP > 4 ? VP : 4) * C * sizeof(float);
    hipLaunchKernelGGL((colreduce_kernel<T, 2, InBwdOp<T>>), dim3(p.nchunks, N), dim3(256), lds, st, op, (int)V, C, p.chunk_vox,
                       partial);
    fin_launch(st, (const float*)partial, N, p.nchunks, 2, C,
                       (double)V, 0.f, (int)FIN_MEAN2, m12);
    int G = sweep_grid(V * CV, CV);
    const T* outp = use_mask ? (const T*)out->ptr : nullptr;
    int ldo = use_mask ? out->ld : 0;
    if (!d_residual)
      RX_LAUNCH_APPLY(false, false);
    else if (accumulate_residual)
      RX_LAUNCH_APPLY(true, true);
    else
      RX_LAUNCH_APPLY(true, false);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_bwd");
  return RX_OK;
}


// ---- residual-block epilogue backward with the MASKED gradient materialised once --------------------------------------
// out = lrelu(IN(y) + res) (resblocks.py:113-114).  The masked gradient g' = g * lrelu'(out) is BOTH the input of the
// InstanceNorm backward and the gradient of the residual.  rx_instnorm_act_bwd read (g, y, out) twice and wrote dy and
// d_residual in its second pass: 3R + 3R 2W = 8 tensor passes.  Here the reduce pass writes g' into the residual-gradient
// buffer while it accumulates sum g' / sum g'*xhat (3R 1W) and the apply pass reads only (g', y) and writes dy (2R 1W): 7
// passes, and the apply kernel is the mask-free, residual-free instantiation.  pool_dy (optional): the gradient of the
// AvgPool that opens the NEXT stage's skip path (resblocks.py:95) -- g is then old_g + pool_dy[v / f] / |f| formed on the fly
// (the separate avgpool_bwd pass over the full-resolution gradient, 1R 1W, disappears as well).
// Sums are taken of g' AS STORED (rounded to the compute type): the apply pass sees exactly the values that were summed.
template <typename T, bool POOL>
struct InBwdResOp {
  ActView<T> g, y, out, pool;
  T* gp;            // masked gradient out (= d_residual), same geometry as y
  long gp_ss;
  int gp_ld;
  const float* stats;
  int C;
  float slope;
  int Yi, Xi, Yo, Xo, fz, fy, fx;
  float inv;
  float mean[Elem<T>::PER16], rstd[Elem<T>::PER16];
  __device__ inline void prepare(int n, int c0) {
#pragma unroll
    for (int j = 0; j < Elem<T>::PER16; ++j) {
      mean[j] = stats[2 * ((size_t)n * C + c0 + j)];
      rstd[j] = stats[2 * ((size_t)n * C + c0 + j) + 1];
    }
  }
  __device__ inline void accumulate(int n, int v, int c0, float (&acc)[2][Elem<T>::PER16]) const {
    constexpr int P = Elem<T>::PER16;
    Vec16<T> gv = ld16(g.at(n, v, c0));
    Vec16<T> yv = ld16(y.at(n, v, c0));
    Vec16<T> ov = ld16(out.at(n, v, c0));
    Vec16<T> pv;
    if (POOL) {
      const int xi = v % Xi, t = v / Xi;
      const int yi = t % Yi, zi = t / Yi;
      const int vo = ((zi / fz) * Yo + (yi / fy)) * Xo + (xi / fx);
      pv = ld16(pool.at(n, vo, c0));
    }
    Vec16<T> w;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(gv.v[j]);
      if (POOL) gg += Elem<T>::to_f(pv.v[j]) * inv;
      if (!(Elem<T>::to_f(ov.v[j]) > 0.f)) gg *= slope;
      w.v[j] = Elem<T>::from_f(gg);
      gg = Elem<T>::to_f(w.v[j]);
      const float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      acc[0][j] += gg;
      acc[1][j] += gg * xh;
    }
    st16(gp + n * gp_ss + (long)v * gp_ld + c0, w);
  }
};

static int check_pool(const rx_act* big, const rx_act* small, const int32_t f[3], const char* who);

extern "C" int rx_instnorm_act_bwd_res(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out, float slope,
                                       const rx_act* pool_dy, const int32_t pool_stride[3], const rx_act* d_residual, const rx_act* dy,
                                       void* ws, size_t ws_bytes, void* stream) {
  static const int32_t one3[3] = {1, 1, 1};
  if (!pool_stride) pool_stride = one3;
  RX_RECORD(stream, [=, g_ = RxActV(g), y_ = RxActV(y), out_ = RxActV(out), pool_dy_ = RxActV(pool_dy), pool_stride_ = RxI3V(pool_stride), d_residual_ = RxActV(d_residual), dy_ = RxActV(dy)](void* s) { return rx_instnorm_act_bwd_res(dt, g_.p(), y_.p(), stats, out_.p(), slope, pool_dy_.p(), pool_stride_.v, d_residual_.p(), dy_.p(), ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(g, dt, "rx_instnorm_act_bwd_res(g)"))) return rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_act_bwd_res(y)"))) return rc;
  if ((rc = check_vec_channels(out, dt, "rx_instnorm_act_bwd_res(out)"))) return rc;
  if ((rc = check_vec_channels(dy, dt, "rx_instnorm_act_bwd_res(dy)"))) return rc;
  if ((rc = check_vec_channels(d_residual, dt, "rx_instnorm_act_bwd_res(d_residual)"))) return rc;
  if (!stats || !ws || !same_geom(y, g) || !same_geom(y, dy) || !same_geom(y, out) || !same_geom(y, d_residual))
    RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_res: bad arguments");
  if (d_residual->ptr == dy->ptr) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_res: d_residual and dy must be different buffers");
  if (pool_dy) {
    if ((rc = check_vec_channels(pool_dy, dt, "rx_instnorm_act_bwd_res(pool_dy)"))) return rc;
    if ((rc = check_pool(y, pool_dy, pool_stride, "rx_instnorm_act_bwd_res"))) return rc;
  }
  const long V = rx_act_voxels(y);
  const int N = y->n, C = y->c;
  size_t need = rx_reduce_ws_bytes(N, V, C, 2) + (size_t)N * C * 2 * sizeof(float);
  if (ws_bytes < need) RX_FAIL(RX_EWORKSPACE, "rx_instnorm_act_bwd_res: workspace too small (%zu < %zu)", ws_bytes, need);
  float* partial = (float*)ws;
  float* m12 = (float*)((char*)ws + rx_align_up(rx_reduce_ws_bytes(N, V, C, 2) - 256, 256));
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, C, P);
    int CV = C / P, VP = 256 / CV;
    size_t lds = (size_t)2 * (VP > 4 ? VP : 4) * C * sizeof(float);
    if (pool_dy) {
      InBwdResOp<T, true> op{make_view<T>(g), make_view<T>(y), make_view<T>(out), make_view<T>(pool_dy), (T*)d_residual->ptr,
                             V * (long)d_residual->ld, d_residual->ld, stats, C, slope, y->y, y->x, pool_dy->y, pool_dy->x,
                             pool_stride[0], pool_stride[1], pool_stride[2], 1.f / (float)(pool_stride[0] * pool_stride[1] * pool_stride[2]), {}, {}};
      hipLaunchKernelGGL((colreduce_kernel<T, 2, InBwdResOp<T, true>>), dim3(p.nchunks, N), dim3(256), lds, st, op, (int)V, C, p.chunk_vox, partial);
    } else {
      InBwdResOp<T, false> op{make_view<T>(g), make_view<T>(y), make_view<T>(out), make_view<T>(y), (T*)d_residual->ptr,
                              V * (long)d_residual->ld, d_residual->ld, stats, C, slope, y->y, y->x, 1, 1, 1, 1, 1, 1.f, {}, {}};
      hipLaunchKernelGGL((colreduce_kernel<T, 2, InBwdResOp<T, false>>), dim3(p.nchunks, N), dim3(256), lds, st, op, (int)V, C, p.chunk_vox, partial);
    }
    fin_launch(st, (const float*)partial, N, p.nchunks, 2, C, (double)V, 0.f, (int)FIN_MEAN2, m12);
    int G = sweep_grid(V * CV, CV);
    // apply: dy = rstd * (g' - m1 - xhat * m2) from (g', y) alone -- no mask, no residual output
    hipLaunchKernelGGL((in_act_bwd_apply_kernel<T, false, false>), dim3(G, N), dim3(256), 0, st, (const T*)d_residual->ptr, d_residual->ld,
                       V * d_residual->ld, (const T*)y->ptr, y->ld, V * y->ld, (const T*)nullptr, 0, 0L, stats, (const float*)m12,
                       (T*)dy->ptr, dy->ld, V * dy->ld, (T*)nullptr, 0, 0L, (int)V, C, slope, 0);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_bwd_res");
  return RX_OK;
}

// ---- InstanceNorm backward with the two means supplied by the caller ------------------------------------------------
// (rx_conv3d_bwd_data_instats: the persistent backward-data kernel accumulates sum g' and sum g'*(y - mean) in its epilogue)
__global__ __launch_bounds__(256) void inbwd_fused_finalize(const float* __restrict__ partial, int N, int nchunks, int C, double V,
                                                            const float* __restrict__ stats, float* __restrict__ m12) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double s0 = 0.0, s1 = 0.0;
  for (int k = lane; k < nchunks; k += 64) {
    const float* p = partial + ((size_t)(n * nchunks + k) * 2) * C + c;
    s0 += (double)p[0];
    s1 += (double)p[C];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (lane != 0) return;
  m12[2 * i] = (float)(s0 / V);
  m12[2 * i + 1] = (float)((double)stats[2 * i + 1] * s1 / V);     // sum g'*xhat = rstd * sum g'*(y - mean)
}
void rx_inbwd_fused_finalize_launch(const float* partial, int N, int nchunks, int C, double V, const float* stats, float* m12, hipStream_t st) {
  hipLaunchKernelGGL(inbwd_fused_finalize, dim3((N * C + 3) / 4), dim3(256), 0, st, partial, N, nchunks, C, V, stats, m12);
}

extern "C" int rx_instnorm_act_bwd_apply(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out, float slope,
                                         const float* m12, const rx_act* dy, const rx_act* d_residual, int accumulate_residual,
                                         void* stream) {
  RX_RECORD(stream, [=, g_ = RxActV(g), y_ = RxActV(y), out_ = RxActV(out), dy_ = RxActV(dy), d_residual_ = RxActV(d_residual)](void* s) { return rx_instnorm_act_bwd_apply(dt, g_.p(), y_.p(), stats, out_.p(), slope, m12, dy_.p(), d_residual_.p(), accumulate_residual, s); });
  int rc;
  if ((rc = check_vec_channels(g, dt, "rx_instnorm_act_bwd_apply(g)"))) return rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_act_bwd_apply(y)"))) return rc;
  if ((rc = check_vec_channels(dy, dt, "rx_instnorm_act_bwd_apply(dy)"))) return rc;
  const bool mask_xhat = slope != 1.0f && out == nullptr;
  const bool use_mask = slope != 1.0f && out != nullptr;
  if (use_mask) {
    if ((rc = check_vec_channels(out, dt, "rx_instnorm_act_bwd_apply(out)"))) return rc;
    if (!same_geom(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_apply: out geometry mismatch");
  }
  if (d_residual) {
    if ((rc = check_vec_channels(d_residual, dt, "rx_instnorm_act_bwd_apply(d_residual)"))) return rc;
    if (!same_geom(y, d_residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_apply: d_residual geometry mismatch");
  }
  if (!stats || !m12 || !same_geom(y, g) || !same_geom(y, dy)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_apply: bad arguments");
  const long V = rx_act_voxels(y);
  const int N = y->n, C = y->c;
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    int CV = C / P;
    int G = sweep_grid(V * CV, CV);
    const T* outp = use_mask ? (const T*)out->ptr : nullptr;
    int ldo = use_mask ? out->ld : 0;
    if (!d_residual)
      RX_LAUNCH_APPLY(false, false);
    else if (accumulate_residual)
      RX_LAUNCH_APPLY(true, true);
    else
      RX_LAUNCH_APPLY(true, false);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_bwd_apply");
  return RX_OK;
}

extern "C" int rx_instnorm_fwd(rx_dtype dt, const rx_act* y, float eps, float* stats, const rx_act* residual, const rx_act* out,
                               float slope, void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), residual_ = RxActV(residual), out_ = RxActV(out)](void* s) { return rx_instnorm_fwd(dt, y_.p(), eps, stats, residual_.p(), out_.p(), slope, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_fwd(y)"))) return rc;
  const long V = rx_act_voxels(y);
  if (V > RX_IN_SMALL_MAX_VOXELS || y->c % 32) {   // large tensors: bandwidth-bound three-launch path
    if ((rc = rx_instnorm_stats(dt, y, eps, stats, ws, ws_bytes, stream))) return rc;
    return rx_instnorm_act_fwd(dt, y, stats, residual, out, slope, stream);
  }
  if ((rc = check_vec_channels(out, dt, "rx_instnorm_fwd(out)"))) return rc;
  if (!stats || !same_geom(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_fwd: geometry mismatch / null stats");
  if (residual) {
    if ((rc = check_vec_channels(residual, dt, "rx_instnorm_fwd(residual)"))) return rc;
    if (!same_geom(y, residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_fwd: residual geometry mismatch");
  }
  hipStream_t st = (hipStream_t)stream;
  const bool narrow = dt != RX_F32 && V >= rx_in_small_narrow();   // (fp32 = parity mode: the summation order the goldens' seeds were screened with)
  dim3 grid(narrow ? y->c / 8 : y->c / 32, y->n);
#define RX_LAUNCH_IN_SMALL_FWD(G, RES)                                                                                              \
  hipLaunchKernelGGL((in_small_fwd_kernel<T, G, RES>), grid, dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld,                  \
                     RES ? (const T*)residual->ptr : (const T*)nullptr, RES ? residual->ld : 0, RES ? V * residual->ld : 0L,         \
                     (T*)out->ptr, out->ld, V * out->ld, stats, (int)V, y->c, eps, slope)
  RX_DISPATCH_DTYPE(dt, T, {
    if (residual && narrow) RX_LAUNCH_IN_SMALL_FWD(8, true);
    else if (residual) RX_LAUNCH_IN_SMALL_FWD(32, true);
    else if (narrow) RX_LAUNCH_IN_SMALL_FWD(8, false);
    else RX_LAUNCH_IN_SMALL_FWD(32, false);
  });
  RX_CHECK_LAUNCH("rx_instnorm_fwd");
  return RX_OK;
}

// ---- AvgPool (kernel = stride, per axis 1 or 2) ---------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, int ldx, long sx, T* __restrict__ y, int ldy, long sy,
                                                          int Zo, int Yo, int Xo, int Yi, int Xi, int C, int fz, int fy, int fx) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)Zo * Yo * Xo * CV;
  const float inv = 1.f / (float)(fz * fy * fx);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cv = (int)(i % CV);
    long vo = i / CV;
    int xo = (int)(vo % Xo);
    int yo = (int)((vo / Xo) % Yo);
    int zo = (int)(vo / ((long)Xo * Yo));
    float acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = 0.f;
    for (int a = 0; a < fz; ++a)
      for (int b = 0; b < fy; ++b)
        for (int c = 0; c < fx; ++c) {
          long vi = ((long)(zo * fz + a) * Yi + (yo * fy + b)) * Xi + (xo * fx + c);
          Vec16<T> t = ld16(x + n * sx + vi * ldx + cv * P);
#pragma unroll
          for (int j = 0; j < P; ++j) acc[j] += Elem<T>::to_f(t.v[j]);
        }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < P; ++j) o.v[j] = Elem<T>::from_f(acc[j] * inv);
    st16(y + n * sy + vo * ldy + cv * P, o);
  }
}

template <typename T, bool ACC>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, int ldy, long sy, T* __restrict__ dx, int ldx, long sx,
                                                          int Zi, int Yi, int Xi, int Yo, int Xo, int C, int fz, int fy, int fx) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)Zi * Yi * Xi * CV;
  const float inv = 1.f / (float)(fz * fy * fx);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cv = (int)(i % CV);
    long vi = i / CV;
    int xi = (int)(vi % Xi);
    int yi = (int)((vi / Xi) % Yi);
    int zi = (int)(vi / ((long)Xi * Yi));
    long vo = ((long)(zi / fz) * Yo + (yi / fy)) * Xo + (xi / fx);
    Vec16<T> t = ld16(dy + n * sy + vo * ldy + cv * P);
    Vec16<T> o;
    if (ACC) o = ld16(dx + n * sx + vi * ldx + cv * P);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = Elem<T>::to_f(t.v[j]) * inv;
      if (ACC) f += Elem<T>::to_f(o.v[j]);
      o.v[j] = Elem<T>::from_f(f);
    }
    st16(dx + n * sx + vi * ldx + cv * P, o);
  }
}

static int check_pool(const rx_act* big, const rx_act* small, const int32_t f[3], const char* who) {
  for (int i = 0; i < 3; ++i)
    if (f[i] < 1 || f[i] > RX_MAX_STRIDE) RX_FAIL(RX_EUNSUPPORTED, "%s: pool factor must be 1..%d per axis", who, RX_MAX_STRIDE);
  if (big->n != small->n || big->c != small->c || big->z != small->z * f[0] || big->y != small->y * f[1] || big->x != small->x * f[2])
    RX_FAIL(RX_EINVAL, "%s: geometry mismatch (%d,%d,%d)/(%d,%d,%d)", who, big->z, big->y, big->x, small->z, small->y, small->x);
  return RX_OK;
}

extern "C" int rx_avgpool_fwd(rx_dtype dt, const rx_act* x, const rx_act* y, const int32_t stride[3], void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), y_ = RxActV(y), stride_ = RxI3V(stride)](void* s) { return rx_avgpool_fwd(dt, x_.p(), y_.p(), stride_.v, s); });
  int rc;
  if ((rc = check_vec_channels(x, dt, "rx_avgpool_fwd(x)"))) return rc;
  if ((rc = check_vec_channels(y, dt, "rx_avgpool_fwd(y)"))) return rc;
  if ((rc = check_pool(x, y, stride, "rx_avgpool_fwd"))) return rc;
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    long total = rx_act_voxels(y) * (y->c / P);
    int G = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(G, x->n), dim3(256), 0, st, (const T*)x->ptr, x->ld, rx_act_voxels(x) * x->ld,
                       (T*)y->ptr, y->ld, rx_act_voxels(y) * y->ld, y->z, y->y, y->x, x->y, x->x, x->c, stride[0], stride[1], stride[2]);
  });
  RX_CHECK_LAUNCH("rx_avgpool_fwd");
  return RX_OK;
}

// ---- block epilogue + the AvgPool of the next block's skip path in ONE pass -----------------------------------------------
// out = lrelu((y-mean)*rstd + res) and pooled = avgpool(out): a thread owns one POOLED voxel's channel vector, produces the
// fz*fy*fx outputs under it and averages them as stored (same values, same summation order as avgpool_fwd_kernel reading
// `out` back -- which moves the whole tensor through HBM a second time).
template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void in_act_pool_fwd_kernel(const T* __restrict__ y, int ldy, long sy, const float* __restrict__ stats,
                                                              const T* __restrict__ res, int ldr, long sr, T* __restrict__ out, int ldo, long so,
                                                              T* __restrict__ pooled, int ldp, long sp, int Zo, int Yo, int Xo, int Yi, int Xi,
                                                              int C, int fz, int fy, int fx, float slope) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P, n = blockIdx.y;
  const long total = (long)Zo * Yo * Xo * CV, step = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = stats[2 * ((size_t)n * C + cv * P + j)];
    rstd[j] = stats[2 * ((size_t)n * C + cv * P + j) + 1];
  }
  const float inv = 1.f / (float)(fz * fy * fx);
  for (; i < total; i += step) {
    const long vo = i / CV;
    const int xo = (int)(vo % Xo), yo = (int)((vo / Xo) % Yo), zo = (int)(vo / ((long)Xo * Yo));
    float acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = 0.f;
    for (int a = 0; a < fz; ++a)
      for (int b = 0; b < fy; ++b)
        for (int c = 0; c < fx; ++c) {
          const long vi = ((long)(zo * fz + a) * Yi + (yo * fy + b)) * Xi + (xo * fx + c);
          Vec16<T> t = ld16(y + n * sy + vi * ldy + cv * P), r, o;
          if (HAS_RES) r = ld16(res + n * sr + vi * ldr + cv * P);
#pragma unroll
          for (int j = 0; j < P; ++j) {
            float f = (Elem<T>::to_f(t.v[j]) - mean[j]) * rstd[j];
            if (HAS_RES) f += Elem<T>::to_f(r.v[j]);
            f = f > 0.f ? f : f * slope;
            o.v[j] = Elem<T>::from_f(f);
            acc[j] += Elem<T>::to_f(o.v[j]);
          }
          st16(out + n * so + vi * ldo + cv * P, o);
        }
    Vec16<T> p;
#pragma unroll
    for (int j = 0; j < P; ++j) p.v[j] = Elem<T>::from_f(acc[j] * inv);
    st16(pooled + n * sp + vo * ldp + cv * P, p);
  }
}

extern "C" int rx_instnorm_act_pool_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* residual, const rx_act* out,
                                        const rx_act* pooled, const int32_t stride[3], float slope, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), residual_ = RxActV(residual), out_ = RxActV(out), pooled_ = RxActV(pooled), stride_ = RxI3V(stride)](void* s) { return rx_instnorm_act_pool_fwd(dt, y_.p(), stats, residual_.p(), out_.p(), pooled_.p(), stride_.v, slope, s); });
  int rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_act_pool_fwd(y)"))) return rc;
  if ((rc = check_vec_channels(out, dt, "rx_instnorm_act_pool_fwd(out)"))) return rc;
  if ((rc = check_vec_channels(pooled, dt, "rx_instnorm_act_pool_fwd(pooled)"))) return rc;
  if (!stats || !same_geom(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_pool_fwd: geometry mismatch / null stats");
  if ((rc = check_pool(out, pooled, stride, "rx_instnorm_act_pool_fwd"))) return rc;
  if (residual) {
    if ((rc = check_vec_channels(residual, dt, "rx_instnorm_act_pool_fwd(residual)"))) return rc;
    if (!same_geom(y, residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_pool_fwd: residual geometry mismatch");
  }
  const long V = rx_act_voxels(y), Vp = rx_act_voxels(pooled);
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    const int CV = y->c / P;
    const int G = sweep_grid(Vp * CV * 4, CV);      // a thread produces up to 8 outputs: 4x the blocks of a plain sweep of Vp
    if (residual)
      hipLaunchKernelGGL((in_act_pool_fwd_kernel<T, true>), dim3(G, y->n), dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld, stats,
                         (const T*)residual->ptr, residual->ld, V * residual->ld, (T*)out->ptr, out->ld, V * out->ld, (T*)pooled->ptr,
                         pooled->ld, Vp * pooled->ld, pooled->z, pooled->y, pooled->x, y->y, y->x, y->c, stride[0], stride[1], stride[2],
                         slope);
    else
      hipLaunchKernelGGL((in_act_pool_fwd_kernel<T, false>), dim3(G, y->n), dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld, stats,
                         (const T*)nullptr, 0, 0L, (T*)out->ptr, out->ld, V * out->ld, (T*)pooled->ptr, pooled->ld, Vp * pooled->ld,
                         pooled->z, pooled->y, pooled->x, y->y, y->x, y->c, stride[0], stride[1], stride[2], slope);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_pool_fwd");
  return RX_OK;
}

extern "C" int rx_avgpool_bwd(rx_dtype dt, const rx_act* dy, const rx_act* dx, const int32_t stride[3], int accumulate,
                              void* stream) {
  RX_RECORD(stream, [=, dy_ = RxActV(dy), dx_ = RxActV(dx), stride_ = RxI3V(stride)](void* s) { return rx_avgpool_bwd(dt, dy_.p(), dx_.p(), stride_.v, accumulate, s); });
  int rc;
  if ((rc = check_vec_channels(dx, dt, "rx_avgpool_bwd(dx)"))) return rc;
  if ((rc = check_vec_channels(dy, dt, "rx_avgpool_bwd(dy)"))) return rc;
  if ((rc = check_pool(dx, dy, stride, "rx_avgpool_bwd"))) return rc;
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    long total = rx_act_voxels(dx) * (dx->c / P);
    int G = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (accumulate)
      hipLaunchKernelGGL((avgpool_bwd_kernel<T, true>), dim3(G, dx->n), dim3(256), 0, st, (const T*)dy->ptr, dy->ld,
                         rx_act_voxels(dy) * dy->ld, (T*)dx->ptr, dx->ld, rx_act_voxels(dx) * dx->ld, dx->z, dx->y, dx->x, dy->y,
                         dy->x, dx->c, stride[0], stride[1], stride[2]);
    else
      hipLaunchKernelGGL((avgpool_bwd_kernel<T, false>), dim3(G, dx->n), dim3(256), 0, st, (const T*)dy->ptr, dy->ld,
                         rx_act_voxels(dy) * dy->ld, (T*)dx->ptr, dx->ld, rx_act_voxels(dx) * dx->ld, dx->z, dx->y, dx->x, dy->y,
                         dy->x, dx->c, stride[0], stride[1], stride[2]);
  });
  RX_CHECK_LAUNCH("rx_avgpool_bwd");
  return RX_OK;
}

// ---- task head: 1x1x1 conv with bias (forward: accumulator arrays sized 8 / 16 / 32 / 64, K <= 8 keeps the lean kernel, more
// than 64 classes run in chunks of 64 with the eval-mode softmax as a separate pass over the logits; backward: weight / bias
// gradients in chunks of <= 16 output channels, the data gradient over all K in chunk 0).  decoder.py:131 puts no bound on
// num_classes (whole-body label sets have 100+); RX_HEAD_MAXK only bounds the LDS-resident weight table of the backward. ---
#define RX_HEAD_MAXK 1024
template <typename T, int MAXK>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int ldx, long sx, const float* __restrict__ w,
                                                       const float* __restrict__ b, int K, int k0, int Kt, float* __restrict__ out, int V,
                                                       int C, int act) {
  constexpr int P = Elem<T>::PER16;
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [K][C]: output channels [k0, k0 + K) of a head with Kt of them
  w += (size_t)k0 * C;
  b += k0;
  for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const int CV = C / P;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    float acc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[k] = k < K ? b[k] : 0.f;
    const T* xp = x + n * sx + v * ldx;
    for (int cv = 0; cv < CV; ++cv) {
      Vec16<T> t = ld16(xp + cv * P);
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < K) {
#pragma unroll
          for (int j = 0; j < P; ++j) acc[k] += Elem<T>::to_f(t.v[j]) * sw[k * C + cv * P + j];
        }
    }
    if (act == RX_ACT_SIGMOID) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k) acc[k] = 1.f / (1.f + expf(-acc[k]));
    } else if (act == RX_ACT_SOFTMAX) {
      float m = -INFINITY, s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < K) m = fmaxf(m, acc[k]);
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < K) {
          acc[k] = expf(acc[k] - m);
          s += acc[k];
        }
#pragma unroll
      for (int k = 0; k < MAXK; ++k) acc[k] = acc[k] / s;
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < K) out[((size_t)n * Kt + k0 + k) * V + v] = acc[k];
  }
}

// softmax over the channel axis of (N, K, V) fp32 logits, in place (heads with more than 64 classes in eval mode)
__global__ __launch_bounds__(256) void softmax_ncdhw_kernel(float* __restrict__ out, int K, long V) {
  const int n = blockIdx.y;
  float* o = out + (size_t)n * K * V;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, o[(size_t)k * V + v]);
    for (int k = 0; k < K; ++k) s += expf(o[(size_t)k * V + v] - m);
    const float inv = 1.f / s;
    for (int k = 0; k < K; ++k) o[(size_t)k * V + v] = expf(o[(size_t)k * V + v] - m) * inv;
  }
}

extern "C" int rx_head_fwd(rx_dtype dt, const rx_act* x, const float* w, const float* b, int k, float* out_ncdhw, int act,
                           void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x)](void* s) { return rx_head_fwd(dt, x_.p(), w, b, k, out_ncdhw, act, s); });
  int rc;
  if ((rc = check_vec_channels(x, dt, "rx_head_fwd"))) return rc;
  if (!w || !b || !out_ncdhw) RX_FAIL(RX_EINVAL, "rx_head_fwd: null pointer");
  if (k < 1 || k > RX_HEAD_MAXK) RX_FAIL(RX_EUNSUPPORTED, "rx_head_fwd: 1 <= K <= %d (got %d)", RX_HEAD_MAXK, k);
  const long V = rx_act_voxels(x);
  hipStream_t st = (hipStream_t)stream;
#define RX_LAUNCH_HEAD_FWD(MK)                                                                                                        \
  hipLaunchKernelGGL((head_fwd_kernel<T, MK>), dim3(G, x->n), dim3(256), (size_t)kc * x->c * sizeof(float), st, (const T*)x->ptr, x->ld, \
                     V * x->ld, w, b, kc, k0, k, out_ncdhw, (int)V, x->c, act_here)
  RX_DISPATCH_DTYPE(dt, T, {
    int G = (int)((V + 255) / 256 > 4096 ? 4096 : (V + 255) / 256);
    const int act_here = (k > 64 && act == RX_ACT_SOFTMAX) ? (int)RX_ACT_NONE : act;      // softmax needs every class: second pass
    for (int k0 = 0; k0 < k; k0 += 64) {
      const int kc = k - k0 < 64 ? k - k0 : 64;
      if (kc <= 8)
        RX_LAUNCH_HEAD_FWD(8);
      else if (kc <= 16)
        RX_LAUNCH_HEAD_FWD(16);
      else if (kc <= 32)
        RX_LAUNCH_HEAD_FWD(32);
      else
        RX_LAUNCH_HEAD_FWD(64);
    }
    if (k > 64 && act == RX_ACT_SOFTMAX) hipLaunchKernelGGL(softmax_ncdhw_kernel, dim3(G, x->n), dim3(256), 0, st, out_ncdhw, k, V);
  });
#undef RX_LAUNCH_HEAD_FWD
  RX_CHECK_LAUNCH("rx_head_fwd");
  return RX_OK;
}

// backward: dx[v][c] = sum_k dout[k][v] w[k][c]; dw[k][c] = sum_v dout[k][v] x[v][c]; db[k] = sum_v dout[k][v]
// One launch handles the output channels [k0, k0 + K) of a head with Kt of them: dw / db of that range; dx (over ALL Kt channels,
// a run-time loop: it needs no per-channel registers) when dx != nullptr -- the caller passes it with the first chunk only.
template <typename T, int MAXK>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ x, int ldx, long sx,
                                                       const float* __restrict__ w, int K, int k0, int Kt, T* __restrict__ dx, int lddx,
                                                       long sdx, int V, int C, int chunk_vox,
                                                       float* __restrict__ partial /*[N][nch][K+1][C]*/) {
  constexpr int P = Elem<T>::PER16;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // sw[Kt][C] then red[(K+1)][VP][C]
  float* sw = sm;
  const int CV = C / P;
  const int VP = 256 / CV > 0 ? 256 / CV : 1;
  float* red = sm + Kt * C;
  for (int i = threadIdx.x; i < Kt * C; i += 256) sw[i] = w[i];
  __syncthreads();
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int vl = tid / CV, cv = tid - vl * CV;
  float aw[MAXK][P];
  float ab[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < P; ++j) aw[k][j] = 0.f;
  }
  const int v_begin = chunk * chunk_vox, v_end = min(V, v_begin + chunk_vox);
  if (vl < VP) {
    for (int v = v_begin + vl; v < v_end; v += VP) {
      Vec16<T> xv = ld16(x + n * sx + (long)v * ldx + cv * P);
      float d[P];
#pragma unroll
      for (int j = 0; j < P; ++j) d[j] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < K) {
          float gk = dout[((size_t)n * Kt + k0 + k) * V + v];
          ab[k] += gk;
#pragma unroll
          for (int j = 0; j < P; ++j) {
            aw[k][j] += gk * Elem<T>::to_f(xv.v[j]);
            if (Kt == K) d[j] += gk * sw[k * C + cv * P + j];
          }
        }
      if (dx && Kt != K) {      // more channels than this launch's chunk: the data gradient sums over all of them (k ascending)
        for (int k = 0; k < Kt; ++k) {
          const float gk = dout[((size_t)n * Kt + k) * V + v];
#pragma unroll
          for (int j = 0; j < P; ++j) d[j] += gk * sw[k * C + cv * P + j];
        }
      }
      if (dx) {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < P; ++j) o.v[j] = Elem<T>::from_f(d[j]);
        st16(dx + n * sdx + (long)v * lddx + cv * P, o);
      }
    }
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < P; ++j) red[(k * VP + vl) * C + cv * P + j] = aw[k][j];
    // bias plane: only column cv==0 carries the sum, stored at channel 0 of plane K
    if (cv == 0)
      for (int k = 0; k < K; ++k) red[(K * VP + vl) * C + k] = ab[k];
  }
  __syncthreads();
  float* pout = partial + (size_t)(n * gridDim.x + chunk) * (K + 1) * C;
  for (int i = tid; i < (K + 1) * C; i += 256) {
    int a = i / C, c = i - a * C;
    if (a == K && c >= K) {
      pout[i] = 0.f;
      continue;
    }
    float s = 0.f;
    for (int r = 0; r < VP; ++r) s += red[(a * VP + r) * C + c];
    pout[i] = s;
  }
}

extern "C" size_t rx_head_bwd_workspace(const rx_act* x, int k) {
  if (!rx_act_ok(x)) return 0;
  return rx_reduce_ws_bytes(x->n, rx_act_voxels(x), x->c, k + 1) + (size_t)(k + 1) * x->c * sizeof(float) + 256;
}

extern "C" int rx_head_bwd(rx_dtype dt, const float* dout_ncdhw, const rx_act* x, const float* w, int k, const rx_act* dx, float* dw,
                           float* db, void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), dx_ = RxActV(dx)](void* s) { return rx_head_bwd(dt, dout_ncdhw, x_.p(), w, k, dx_.p(), dw, db, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(x, dt, "rx_head_bwd(x)"))) return rc;
  if (dx) {
    if ((rc = check_vec_channels(dx, dt, "rx_head_bwd(dx)"))) return rc;
    if (!same_geom(x, dx)) RX_FAIL(RX_EINVAL, "rx_head_bwd: dx geometry mismatch");
  }
  if (!dout_ncdhw || !w || !dw || !db || !ws) RX_FAIL(RX_EINVAL, "rx_head_bwd: null pointer");
  if (k < 1 || k > RX_HEAD_MAXK || (k <= 16 && k > x->c) || x->c < 16) RX_FAIL(RX_EUNSUPPORTED, "rx_head_bwd: 1 <= K <= %d", RX_HEAD_MAXK);
  const long V = rx_act_voxels(x);
  const int N = x->n, C = x->c;
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, C, P);
    int kc_max = k <= 16 ? k : 16;                 // output channels per launch
    {   // many classes: the LDS-resident weight table grows with K -- halve the chunk until table + reduction planes fit
      const int VP0 = 256 / (C / P);
      while (kc_max > 4 && ((size_t)k * C + (size_t)(kc_max + 1) * VP0 * C) * sizeof(float) > 150 * 1024) kc_max /= 2;
    }
    size_t need = (size_t)N * p.nchunks * (kc_max + 1) * C * sizeof(float) + (size_t)(kc_max + 1) * C * sizeof(float) + 256;
    if (ws_bytes < need) RX_FAIL(RX_EWORKSPACE, "rx_head_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    int CV = C / P, VP = 256 / CV;
    float* partial = (float*)ws;
    float* fin = partial + (size_t)N * p.nchunks * (kc_max + 1) * C;
    size_t lds = ((size_t)k * C + (size_t)(kc_max + 1) * VP * C) * sizeof(float);
    if (lds > 160 * 1024) RX_FAIL(RX_EUNSUPPORTED, "rx_head_bwd: K = %d needs %zu bytes of LDS", k, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<T, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<T, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int k0 = 0; k0 < k; k0 += kc_max) {
      const int kc = k - k0 < kc_max ? k - k0 : kc_max;
      T* dxp = (dx && k0 == 0) ? (T*)dx->ptr : (T*)nullptr;
      if (kc <= 8) {
        hipLaunchKernelGGL((head_bwd_kernel<T, 8>), dim3(p.nchunks, N), dim3(256), lds, st, dout_ncdhw, (const T*)x->ptr, x->ld, V * x->ld, w, kc,
                           k0, k, dxp, dx ? dx->ld : 0, dx ? V * dx->ld : 0L, (int)V, C, p.chunk_vox, partial);
      } else {
        hipLaunchKernelGGL((head_bwd_kernel<T, 16>), dim3(p.nchunks, N), dim3(256), lds, st, dout_ncdhw, (const T*)x->ptr, x->ld, V * x->ld, w, kc,
                           k0, k, dxp, dx ? dx->ld : 0, dx ? V * dx->ld : 0L, (int)V, C, p.chunk_vox, partial);
      }
      fin_launch(st, (const float*)partial, N, p.nchunks, kc + 1,
                         C, (double)V, 0.f, (int)FIN_SUM_OVER_N, fin);
      (void)hipMemcpyAsync(dw + (size_t)k0 * C, fin, (size_t)kc * C * sizeof(float), hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(db + k0, fin + (size_t)kc * C, (size_t)kc * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
  });
  RX_CHECK_LAUNCH("rx_head_bwd");
  return RX_OK;
}

#define RX_HEADG_MAXK 4
// ---- InstanceNorm + LeakyReLU of the layer under a task head, with the head's 1x1x1 conv in the same pass -----------
// rx_head_fwd re-read the activated output (268 MB at cfg2) to form K logits per voxel.  Here the CV lanes that hold one voxel's
// channel vectors pass the running sums along (lane cv adds the partial dot product of its 8 channels to what lane cv-1 holds, from
// the rounded output values; logits agree with head_fwd_kernel's sequential sum to fp32 round-off); the last lane
// applies the eval-mode activation and writes the NCDHW fp32 logits.  K <= 4, no residual (decoder.py:115-131).
template <typename T>
__global__ __launch_bounds__(256) void in_act_head_fwd_kernel(const T* __restrict__ y, int ldy, long sy, const float* __restrict__ stats,
                                                              T* __restrict__ out, int ldo, long so, int V, int C, float slope,
                                                              const float* __restrict__ hw, const float* __restrict__ hb, int K,
                                                              float* __restrict__ logits, int act) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)V * CV;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P], w[RX_HEADG_MAXK][P], b[RX_HEADG_MAXK];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = stats[2 * ((size_t)n * C + cv * P + j)];
    rstd[j] = stats[2 * ((size_t)n * C + cv * P + j) + 1];
#pragma unroll
    for (int k = 0; k < RX_HEADG_MAXK; ++k) w[k][j] = k < K ? hw[k * C + cv * P + j] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < RX_HEADG_MAXK; ++k) b[k] = k < K ? hb[k] : 0.f;
  const T* yn = y + n * sy;
  T* on = out ? out + n * so : nullptr;
  for (; i < total; i += step) {
    long v = i / CV;
    Vec16<T> a = ld16(yn + v * ldy + cv * P);
    Vec16<T> o;
    float of[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = (Elem<T>::to_f(a.v[j]) - mean[j]) * rstd[j];
      f = f > 0.f ? f : f * slope;
      o.v[j] = Elem<T>::from_f(f);
      of[j] = Elem<T>::to_f(o.v[j]);
    }
    if (on) st16(on + v * ldo + cv * P, o);      // (out == NULL: nobody reads the activated output -- see rx_instnorm_act_bwd_head's dw / db)
    // The CV lanes of a voxel each form the partial dot products of THEIR 8 channels, then hand a running sum along in channel
    // order (lane cv adds its partial to what lane cv-1 holds).  No divergent region between the cross-lane moves: every lane
    // computes, a select keeps the owner's value.  (The first version did the adds inside `if (cv == s)`; with a SECOND process
    // time-slicing the GPU -- two DDP ranks rehearsed on one device -- a few logits per pass then came out different while every
    // other tensor of the pass stayed bit-identical, with ds_bpermute and with DPP moves alike; 0 of 120 passes with this form,
    // scripts/fwd_layer_diag.py.  Never observed with one process per GPU.)
    float part[RX_HEADG_MAXK], acc[RX_HEADG_MAXK];
#pragma unroll
    for (int k = 0; k < RX_HEADG_MAXK; ++k) {
      part[k] = 0.f, acc[k] = b[k];
      if (k < K) {
#pragma unroll
        for (int j = 0; j < P; ++j) part[k] += of[j] * w[k][j];
      }
    }
    for (int s = 0; s < CV; ++s) {
#pragma unroll
      for (int k = 0; k < RX_HEADG_MAXK; ++k)
        if (k < K) {                     // K is uniform: a scalar branch
          const float prev = __shfl_up(acc[k], 1, 64);
          const float t = (s == 0 ? b[k] : prev) + part[k];
          acc[k] = cv == s ? t : acc[k];
        }
    }
    if (cv == CV - 1) {
      if (act == RX_ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < RX_HEADG_MAXK; ++k) acc[k] = 1.f / (1.f + expf(-acc[k]));
      } else if (act == RX_ACT_SOFTMAX) {
        float m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int k = 0; k < RX_HEADG_MAXK; ++k)
          if (k < K) m = fmaxf(m, acc[k]);
#pragma unroll
        for (int k = 0; k < RX_HEADG_MAXK; ++k)
          if (k < K) {
            acc[k] = expf(acc[k] - m);
            sum += acc[k];
          }
#pragma unroll
        for (int k = 0; k < RX_HEADG_MAXK; ++k) acc[k] = acc[k] / sum;
      }
#pragma unroll
      for (int k = 0; k < RX_HEADG_MAXK; ++k)
        if (k < K) logits[((size_t)n * K + k) * V + v] = acc[k];
    }
  }
}

// out = lrelu((y - mean) * rstd) AND out_ncdhw = head(out) (+ eval-mode activation) in one pass; `stats` = (mean, rstd) of y.
// Same `out` bit for bit and the same logits to fp32 round-off as rx_instnorm_act_fwd followed by rx_head_fwd.  K <= 4, 64 % (C / 8) == 0.
extern "C" int rx_instnorm_act_head_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* out, float slope,
                                        const float* head_w, const float* head_b, int k, float* out_ncdhw, int act, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), out_ = RxActV(out)](void* s) { return rx_instnorm_act_head_fwd(dt, y_.p(), stats, out_.p(), slope, head_w, head_b, k, out_ncdhw, act, s); });
  int rc = check_vec_channels(y, dt, "rx_instnorm_act_head_fwd(y)");
  if (rc) return rc;
  if (out && (rc = check_vec_channels(out, dt, "rx_instnorm_act_head_fwd(out)"))) return rc;
  if (!stats || !head_w || !head_b || !out_ncdhw || (out && !same_geom(y, out))) RX_FAIL(RX_EINVAL, "rx_instnorm_act_head_fwd: bad arguments");
  if (dt == RX_F32 || k < 1 || k > RX_HEADG_MAXK || 64 % (y->c / 8) != 0)
    RX_FAIL(RX_EUNSUPPORTED, "rx_instnorm_act_head_fwd: 16-bit types, K <= %d, C / 8 dividing 64", RX_HEADG_MAXK);
  const long V = rx_act_voxels(y);
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    int CV = y->c / P;
    int G = sweep_grid(V * CV, CV);
    hipLaunchKernelGGL((in_act_head_fwd_kernel<T>), dim3(G, y->n), dim3(256), 0, st, (const T*)y->ptr, y->ld, V * y->ld, stats,
                       out ? (T*)out->ptr : (T*)nullptr, out ? out->ld : 0, out ? V * out->ld : 0L, (int)V, y->c, slope, head_w, head_b, k,
                       out_ncdhw, act);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_head_fwd");
  return RX_OK;
}

// ---- InstanceNorm backward of the layer that feeds a task head, with the head's data gradient formed on the fly ------
// The gradient that reaches the last decoder conv block is rank K: g[v][c] = sum_k dlogit[k][v] * w_head[k][c] (K = 1 for a
// segmentation head, 3 for normals).  rx_head_bwd used to write it as a full (N, V, C) tensor (268 MB at cfg2) that the two
// passes of the InstanceNorm backward then read back twice.  Here both passes rebuild g from the fp32 logit gradient (4*K bytes
// per voxel instead of 2*C) and the head's weights; rx_head_bwd is called with dx = NULL and only reduces dw / db.  g is
// rounded to the storage type exactly where rx_head_bwd rounded it, so dy is bit-identical to the three-tensor path.
// KW > 0 (round 3): the same pass also reduces the HEAD's parameter gradients, dw[k][c] = sum_v dout[k][v] * a[v][c] and db[k] =
// sum_v dout[k][v], with a = lrelu(xhat) recomputed from y and rounded to the storage type as the forward stored it -- the
// activated output of the layer under a head (268 MB at cfg2) is then neither written by the forward nor read by rx_head_bwd
// (a 131 us launch at cfg2), which is not called at all.  Accumulators 2 .. 2+K-1 hold dw, accumulator 2+K holds db[k] in lane k
// (only the threads of channel vector 0 add to it: every channel vector of a voxel sees the same dout).
template <typename T, int KW = 0>
struct InBwdHeadOp {
  ActView<T> y;
  const float* stats;
  const float* dout;  // (N, K, V) fp32
  const float* hw;    // (K, C)
  int C, K, V;
  float slope;
  bool mask_xhat;
  float mean[Elem<T>::PER16], rstd[Elem<T>::PER16], w[RX_HEADG_MAXK][Elem<T>::PER16];
  __device__ inline void prepare(int n, int c0) {
#pragma unroll
    for (int j = 0; j < Elem<T>::PER16; ++j) {
      mean[j] = stats[2 * ((size_t)n * C + c0 + j)];
      rstd[j] = stats[2 * ((size_t)n * C + c0 + j) + 1];
#pragma unroll
      for (int k = 0; k < RX_HEADG_MAXK; ++k) w[k][j] = k < K ? hw[k * C + c0 + j] : 0.f;
    }
  }
  __device__ inline void accumulate(int n, int v, int c0, float (&acc)[KW ? KW + 3 : 2][Elem<T>::PER16]) const {
    constexpr int P = Elem<T>::PER16;
    Vec16<T> yv = ld16(y.at(n, v, c0));
    float d[P], gk[RX_HEADG_MAXK];
#pragma unroll
    for (int j = 0; j < P; ++j) d[j] = 0.f;
#pragma unroll
    for (int k = 0; k < RX_HEADG_MAXK; ++k) {
      gk[k] = 0.f;
      if (k < K) {
        gk[k] = dout[((size_t)n * K + k) * V + v];
#pragma unroll
        for (int j = 0; j < P; ++j) d[j] += gk[k] * w[k][j];
      }
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(Elem<T>::from_f(d[j]));
      float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (mask_xhat && !(xh > 0.f)) gg *= slope;
      acc[0][j] += gg;
      acc[1][j] += gg * xh;
      if (KW) {
        const float a = Elem<T>::to_f(Elem<T>::from_f(xh > 0.f ? xh : xh * slope));      // what rx_instnorm_act_head_fwd stored
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[2 + k][j] += gk[k] * a;
      }
    }
    if (KW && c0 == 0) {
#pragma unroll
      for (int k = 0; k < KW; ++k) acc[2 + KW][k] += gk[k];
    }
  }
};

// finalize of the pass above: m12[n][c] (both means), dw[k][c] and db[k] (sums over n and chunks), one workgroup per output
__global__ __launch_bounds__(256) void inbwd_head_finalize(const float* __restrict__ partial, int N, int nchunks, int nacc, int C, int K, double V,
                                                           float* __restrict__ m12, float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  if (i < N * C) {
    const int n = i / C, c = i - n * C;
    fin_gather(partial + ((size_t)n * nchunks * nacc) * C + c, (size_t)nacc * C, nchunks, C, s0, s1);
    if (threadIdx.x == 0) m12[2 * i] = (float)(s0 / V), m12[2 * i + 1] = (float)(s1 / V);
    return;
  }
  const int j = i - N * C;
  if (j < K * C) {
    const int k = j / C, c = j - k * C;
    fin_gather(partial + (size_t)(2 + k) * C + c, (size_t)nacc * C, N * nchunks, 0, s0, s1);
    if (threadIdx.x == 0) dw[j] = (float)s0;
    return;
  }
  const int k = j - K * C;
  fin_gather(partial + (size_t)(2 + K) * C + k, (size_t)nacc * C, N * nchunks, 0, s0, s1);
  if (threadIdx.x == 0) db[k] = (float)s0;
}

template <typename T>
__global__ __launch_bounds__(256) void in_act_bwd_apply_head_kernel(const float* __restrict__ dout, int K, const float* __restrict__ hw,
                                                                    const T* __restrict__ y, int ldy, long sy,
                                                                    const float* __restrict__ stats, const float* __restrict__ m12,
                                                                    T* __restrict__ dy, int lddy, long sdy, int V, int C, float slope,
                                                                    int mask_xhat) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P;
  const int n = blockIdx.y;
  const long total = (long)V * CV;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P], m1[P], m2[P], w[RX_HEADG_MAXK][P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    size_t k = (size_t)n * C + cv * P + j;
    mean[j] = stats[2 * k];
    rstd[j] = stats[2 * k + 1];
    m1[j] = m12[2 * k];
    m2[j] = m12[2 * k + 1];
#pragma unroll
    for (int q = 0; q < RX_HEADG_MAXK; ++q) w[q][j] = q < K ? hw[q * C + cv * P + j] : 0.f;
  }
  for (; i < total; i += step) {
    long v = i / CV;
    Vec16<T> yv = ld16(y + n * sy + v * ldy + cv * P);
    float d[P];
#pragma unroll
    for (int j = 0; j < P; ++j) d[j] = 0.f;
#pragma unroll
    for (int q = 0; q < RX_HEADG_MAXK; ++q)
      if (q < K) {
        const float gk = dout[((size_t)n * K + q) * V + v];
#pragma unroll
        for (int j = 0; j < P; ++j) d[j] += gk * w[q][j];
      }
    Vec16<T> dv;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(Elem<T>::from_f(d[j]));
      float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (mask_xhat && !(xh > 0.f)) gg *= slope;
      dv.v[j] = Elem<T>::from_f(rstd[j] * (gg - m1[j] - xh * m2[j]));
    }
    st16(dy + n * sdy + v * lddy + cv * P, dv);
  }
}

// dy = InstanceNorm+LeakyReLU backward of a layer WITHOUT residual whose output gradient is the data gradient of a 1x1x1 head:
// g = dout (N,K,Z,Y,X fp32) x head_w (K,C), never materialised.  Same result as rx_head_bwd(dx = g) + rx_instnorm_act_bwd(g, ...,
// out = NULL).  K <= 4.
#define RX_HEAD_REDUCE(KW_)                                                                                                              \
  do {                                                                                                                                   \
    InBwdHeadOp<T, KW_> op{make_view<T>(y), stats, dout_ncdhw, head_w, C, k, (int)V, slope, mask_xhat, {}, {}, {}};                       \
    hipLaunchKernelGGL((colreduce_kernel<T, (KW_ ? KW_ + 3 : 2), InBwdHeadOp<T, KW_>>), dim3(p.nchunks, N), dim3(256), lds, st, op, (int)V, C, \
                       p.chunk_vox, partial);                                                                                            \
  } while (0)
extern "C" int rx_instnorm_act_bwd_head(rx_dtype dt, const float* dout_ncdhw, int k, const float* head_w, const rx_act* y,
                                        const float* stats, float slope, const rx_act* dy, float* head_dw, float* head_db, void* ws,
                                        size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), dy_ = RxActV(dy)](void* s) { return rx_instnorm_act_bwd_head(dt, dout_ncdhw, k, head_w, y_.p(), stats, slope, dy_.p(), head_dw, head_db, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(y, dt, "rx_instnorm_act_bwd_head(y)"))) return rc;
  if ((rc = check_vec_channels(dy, dt, "rx_instnorm_act_bwd_head(dy)"))) return rc;
  if (!dout_ncdhw || !head_w || !stats || !ws || k < 1 || k > RX_HEADG_MAXK || !same_geom(y, dy))
    RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_head: bad arguments (K must be 1..%d)", RX_HEADG_MAXK);
  const long V = rx_act_voxels(y);
  const int N = y->n, C = y->c;
  if (V > 0x7fffffffL) RX_FAIL(RX_EUNSUPPORTED, "rx_instnorm_act_bwd_head: volume too large");
  if ((head_dw == nullptr) != (head_db == nullptr)) RX_FAIL(RX_EINVAL, "rx_instnorm_act_bwd_head: head_dw and head_db come together");
  const int nacc = head_dw ? k + 3 : 2;
  size_t need = rx_reduce_ws_bytes(N, V, C, nacc) + (size_t)N * C * 2 * sizeof(float);
  if (ws_bytes < need) RX_FAIL(RX_EWORKSPACE, "rx_instnorm_act_bwd_head: workspace too small (%zu < %zu)", ws_bytes, need);
  float* partial = (float*)ws;
  float* m12 = (float*)((char*)ws + rx_align_up(rx_reduce_ws_bytes(N, V, C, nacc) - 256, 256));
  hipStream_t st = (hipStream_t)stream;
  const bool mask_xhat = slope != 1.0f;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    ReducePlan p = rx_reduce_plan(V, C, P);
    int CV = C / P, VP = 256 / CV;
    size_t lds = (size_t)nacc * (VP > 4 ? VP : 4) * C * sizeof(float);
    if (!head_dw) {
      RX_HEAD_REDUCE(0);
      fin_launch(st, (const float*)partial, N, p.nchunks, 2, C, (double)V, 0.f, (int)FIN_MEAN2, m12);
    } else {
      if (k == 1) RX_HEAD_REDUCE(1);
      else if (k == 2) RX_HEAD_REDUCE(2);
      else if (k == 3) RX_HEAD_REDUCE(3);
      else RX_HEAD_REDUCE(4);
      hipLaunchKernelGGL(inbwd_head_finalize, dim3(N * C + k * C + k), dim3(256), 0, st, (const float*)partial, N, p.nchunks, nacc, C, k,
                         (double)V, m12, head_dw, head_db);
    }
    int G = sweep_grid(V * CV, CV);
    hipLaunchKernelGGL((in_act_bwd_apply_head_kernel<T>), dim3(G, N), dim3(256), 0, st, dout_ncdhw, k, head_w, (const T*)y->ptr, y->ld,
                       V * y->ld, stats, (const float*)m12, (T*)dy->ptr, dy->ld, V * dy->ld, (int)V, C, slope, mask_xhat ? 1 : 0);
  });
  RX_CHECK_LAUNCH("rx_instnorm_act_bwd_head");
  return RX_OK;
}
#undef RX_HEAD_REDUCE

// ---- stem convolution on the NCDHW fp32 image (Cin <= 16; the MFMA variants of rx_stem_wgrad.hip take Cin <= 4) ----------
// thread -> (voxel, vector of P output channels); weights in LDS as [tap*Cin][Cout]
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, int Cin, int Z, int Y, int X, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out, int ldo, long so, int Co, int kz,
                                                       int ky, int kx) {
  constexpr int P = Elem<T>::PER16;
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cin*T][Co]
  const int TT = kz * ky * kx;
  for (int i = threadIdx.x; i < Co * Cin * TT; i += 256) {
    int co = i / (Cin * TT), r = i - co * (Cin * TT);  // r = ci*TT + t  (torch layout (Co,Ci,T))
    sw[r * Co + co] = w[i];
  }
  __syncthreads();
  const int n = blockIdx.y;
  const int CV = Co / P;
  const long V = (long)Z * Y * X;
  const long total = V * CV;
  const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cv = (int)(i % CV);
    long v = i / CV;
    int xx = (int)(v % X), yy = (int)((v / X) % Y), zz = (int)(v / ((long)X * Y));
    float acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = bias ? bias[cv * P + j] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xc = x + ((size_t)n * Cin + ci) * V;
      for (int a = 0; a < kz; ++a) {
        int z2 = zz + a - pz;
        if ((unsigned)z2 >= (unsigned)Z) continue;
        for (int b = 0; b < ky; ++b) {
          int y2 = yy + b - py;
          if ((unsigned)y2 >= (unsigned)Y) continue;
          for (int c = 0; c < kx; ++c) {
            int x2 = xx + c - px;
            if ((unsigned)x2 >= (unsigned)X) continue;
            float xv = xc[((long)z2 * Y + y2) * X + x2];
            const float* wr = sw + ((ci * TT) + (a * ky + b) * kx + c) * Co + cv * P;
#pragma unroll
            for (int j = 0; j < P; ++j) acc[j] += xv * wr[j];
          }
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < P; ++j) o.v[j] = Elem<T>::from_f(acc[j]);
    st16(out + n * so + v * ldo + cv * P, o);
  }
}

// One thread per voxel, 32 output channels at a time (16-bit output types): the taps of the voxel are loaded ONCE into
// registers and every weight comes from LDS as a wave-uniform (broadcast) 16-byte read.  The first version above gave a
// voxel to 4 threads of 8 channels each: 4x the image loads and bounds checks (428 us for the cfg2 stem; this one is
// bound by its 864 FMAs per voxel).
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd32_kernel(const float* __restrict__ x, int Cin, int Z, int Y, int X, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ out, int ldo, long so, int Co,
                                                         int kz, int ky, int kx) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cin*TT][Co]
  const int TT = kz * ky * kx;
  for (int i = threadIdx.x; i < Co * Cin * TT; i += 256) {
    int co = i / (Cin * TT), r = i - co * (Cin * TT);
    sw[r * Co + co] = w[i];
  }
  __syncthreads();
  const int n = blockIdx.y;
  const long V = (long)Z * Y * X;
  const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int xx = (int)(v % X), yy = (int)((v / X) % Y), zz = (int)(v / ((long)X * Y));
    for (int c0 = 0; c0 < Co; c0 += 32) {
      float acc[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = bias ? bias[c0 + j] : 0.f;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xc = x + ((size_t)n * Cin + ci) * V;
        for (int a = 0; a < kz; ++a) {
          const int z2 = zz + a - pz;
          for (int b = 0; b < ky; ++b) {
            const int y2 = yy + b - py;
            for (int c = 0; c < kx; ++c) {
              const int x2 = xx + c - px;
              const bool ok = (unsigned)z2 < (unsigned)Z && (unsigned)y2 < (unsigned)Y && (unsigned)x2 < (unsigned)X;
              const float xv = ok ? xc[((long)z2 * Y + y2) * X + x2] : 0.f;
              const f32x4* wr = reinterpret_cast<const f32x4*>(sw + ((ci * TT) + (a * ky + b) * kx + c) * Co + c0);
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const f32x4 wv = wr[q];           // wave-uniform address: LDS broadcast
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * q + j] += xv * wv[j];
              }
            }
          }
        }
      }
      T* op = out + n * so + v * ldo + c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        T vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[j] = Elem<T>::from_f(acc[8 * q + j]);
        *reinterpret_cast<u32x4*>(op + 8 * q) = *reinterpret_cast<u32x4*>(vals);
      }
    }
  }
}

static int check_kernel13(const int32_t k[3], const char* who) {
  for (int i = 0; i < 3; ++i)
    if (k[i] != 1 && k[i] != 3) RX_FAIL(RX_EUNSUPPORTED, "%s: kernel sizes must be 1 or 3", who);
  return RX_OK;
}

int rx_stem_fwd_mfma_try(rx_dtype dt, const float* x, int n, int cin, int z, int y, int xx, const float* w, const float* bias,
                         const rx_act* out, const int32_t kernel[3], hipStream_t st, float* stat_part, size_t stat_bytes, int* stat_chunks);
static int rx_stem_mfma_on() {
  static int mf = -1;      // RX_STEM_MFMA=0: the VALU kernels
  if (mf < 0) {
    const char* e = getenv("RX_STEM_MFMA");
    mf = e ? atoi(e) : 1;
  }
  return mf;
}

// the stem conv and the InstanceNorm statistics of its output (encoder.py:84 + simple_conv_blocks.py:58-72): one pass on the
// MFMA kernel (the separate statistics pass read the 268 MB output of the cfg2 stem again: 109 us of a 17 ms step), the two
// calls otherwise.  Same statistics either way (sums of the values as stored).
extern "C" int rx_stem_conv_fwd_stats(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x, const float* w,
                                      const float* bias, const rx_act* out, const int32_t kernel[3], float eps, float* stats, void* ws,
                                      size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, out_ = RxActV(out), kernel_ = RxI3V(kernel)](void* s) { return rx_stem_conv_fwd_stats(dt, x_ncdhw, n, cin, z, y, x, w, bias, out_.p(), kernel_.v, eps, stats, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(out, dt, "rx_stem_conv_fwd_stats(out)"))) return rc;
  if ((rc = check_kernel13(kernel, "rx_stem_conv_fwd_stats"))) return rc;
  if (!x_ncdhw || !w || !stats || !ws || cin < 1 || cin > 16) RX_FAIL(RX_EINVAL, "rx_stem_conv_fwd_stats: bad arguments");
  if (out->n != n || out->z != z || out->y != y || out->x != x) RX_FAIL(RX_EINVAL, "rx_stem_conv_fwd_stats: geometry mismatch");
  static int fuse = -1;
  if (fuse < 0) {
    const char* e = getenv("RX_FUSED_STATS");
    fuse = e ? atoi(e) : 1;
  }
  int chunks = 0;
  if (fuse && rx_stem_mfma_on() &&
      rx_stem_fwd_mfma_try(dt, x_ncdhw, n, cin, z, y, x, w, bias, out, kernel, (hipStream_t)stream, (float*)ws, ws_bytes, &chunks) == 1) {
    if (chunks > 0) {
      rx_stats_finalize_launch((const float*)ws, n, chunks, out->c, (double)rx_act_voxels(out), eps, stats, (hipStream_t)stream);
      RX_CHECK_LAUNCH("rx_stem_conv_fwd_stats");
      return RX_OK;
    }
    RX_CHECK_LAUNCH("rx_stem_conv_fwd_stats(mfma)");
    return rx_instnorm_stats(dt, out, eps, stats, ws, ws_bytes, stream);
  }
  if ((rc = rx_stem_conv_fwd(dt, x_ncdhw, n, cin, z, y, x, w, bias, out, kernel, stream))) return rc;
  return rx_instnorm_stats(dt, out, eps, stats, ws, ws_bytes, stream);
}

extern "C" int rx_stem_conv_fwd(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x, const float* w,
                                const float* bias, const rx_act* out, const int32_t kernel[3], void* stream) {
  RX_RECORD(stream, [=, out_ = RxActV(out), kernel_ = RxI3V(kernel)](void* s) { return rx_stem_conv_fwd(dt, x_ncdhw, n, cin, z, y, x, w, bias, out_.p(), kernel_.v, s); });
  int rc;
  if ((rc = check_vec_channels(out, dt, "rx_stem_conv_fwd(out)"))) return rc;
  if ((rc = check_kernel13(kernel, "rx_stem_conv_fwd"))) return rc;
  if (!x_ncdhw || !w || cin < 1 || cin > 16) RX_FAIL(RX_EUNSUPPORTED, "rx_stem_conv_fwd: 1 <= Cin <= 16");
  {      // the VALU kernels keep all weights in LDS: [Cin * taps][Cout] floats
    const size_t wl = (size_t)out->c * cin * kernel[0] * kernel[1] * kernel[2] * sizeof(float);
    if (wl > 160 * 1024) RX_FAIL(RX_EUNSUPPORTED, "rx_stem_conv_fwd: %d x %d channels x %d taps do not fit the LDS", cin, out->c, kernel[0] * kernel[1] * kernel[2]);
    if (wl > 48 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd32_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd32_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
  }
  if (out->n != n || out->z != z || out->y != y || out->x != x) RX_FAIL(RX_EINVAL, "rx_stem_conv_fwd: geometry mismatch");
  hipStream_t st = (hipStream_t)stream;
  const int TT = kernel[0] * kernel[1] * kernel[2];
  if (rx_stem_mfma_on() && rx_stem_fwd_mfma_try(dt, x_ncdhw, n, cin, z, y, x, w, bias, out, kernel, st, nullptr, 0, nullptr) == 1) {
    RX_CHECK_LAUNCH("rx_stem_conv_fwd(mfma)");
    return RX_OK;
  }
  if (dt != RX_F32 && out->c % 32 == 0 && (out->c * 4) % 16 == 0 && !getenv("RX_NO_STEM32")) {   // one thread per voxel x 32 channels
    const long V = rx_act_voxels(out);
    const int G = (int)((V + 255) / 256 > 16384 ? 16384 : (V + 255) / 256);
    const size_t lds = (size_t)out->c * cin * TT * sizeof(float);
    if (dt == RX_BF16)
      hipLaunchKernelGGL((stem_fwd32_kernel<bf16_t>), dim3(G, n), dim3(256), lds, st, x_ncdhw, cin, z, y, x, w, bias, (bf16_t*)out->ptr, out->ld,
                         V * out->ld, out->c, kernel[0], kernel[1], kernel[2]);
    else
      hipLaunchKernelGGL((stem_fwd32_kernel<f16_t>), dim3(G, n), dim3(256), lds, st, x_ncdhw, cin, z, y, x, w, bias, (f16_t*)out->ptr, out->ld,
                         V * out->ld, out->c, kernel[0], kernel[1], kernel[2]);
    RX_CHECK_LAUNCH("rx_stem_conv_fwd(32)");
    return RX_OK;
  }
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    long total = rx_act_voxels(out) * (out->c / P);
    int G = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL((stem_fwd_kernel<T>), dim3(G, n), dim3(256), (size_t)out->c * cin * TT * sizeof(float), st, x_ncdhw, cin, z, y, x, w,
                       bias, (T*)out->ptr, out->ld, rx_act_voxels(out) * out->ld, out->c, kernel[0], kernel[1], kernel[2]);
  });
  RX_CHECK_LAUNCH("rx_stem_conv_fwd");
  return RX_OK;
}

// stem weight gradient: dw[co][ci][t] = sum_{n,v} dy[n][v][co] * x[n][ci][v + t - pad]
// thread -> (voxel lane, vector of 4 output channels); 27 accumulators x 4 channels per input
// channel (grid.z = ci); lanes of equal channel-vector are combined with xor-shuffles, waves
// through LDS, blocks through the partial buffer.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, int Cin, int Z, int Y, int X, const T* __restrict__ dy,
                                                         int ldy, long sy, int Co, int kz, int ky, int kx, int N, int chunk_vox,
                                                         float* __restrict__ partial /*[nch][Cin][27][Co]*/) {
  const int ci = blockIdx.z;
  const int CQ = Co / 4;  // channel quads; requires CQ | 64
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane % CQ, vl = (threadIdx.x) / CQ;  // vl in [0, 256/CQ)
  const int VPB = 256 / CQ;
  const long V = (long)Z * Y * X;
  const long NV = (long)N * V;
  const int TT = kz * ky * kx;
  const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;
  float acc[27][4];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  const long q_begin = (long)blockIdx.x * chunk_vox;
  const long q_end = q_begin + chunk_vox < NV ? q_begin + chunk_vox : NV;
  for (long q = q_begin + vl; q < q_end; q += VPB) {
    int n = (int)(q / V);
    long v = q - (long)n * V;
    int xx = (int)(v % X), yy = (int)((v / X) % Y), zz = (int)(v / ((long)X * Y));
    const T* dp = dy + n * sy + v * ldy + cq * 4;
    float d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = Elem<T>::to_f(dp[j]);
    const float* xc = x + ((size_t)n * Cin + ci) * V;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (a < kz && b < ky && c < kx) {
            int z2 = zz + a - pz, y2 = yy + b - py, x2 = xx + c - px;
            float xv = 0.f;
            if ((unsigned)z2 < (unsigned)Z && (unsigned)y2 < (unsigned)Y && (unsigned)x2 < (unsigned)X)
              xv = xc[((long)z2 * Y + y2) * X + x2];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a * 9 + b * 3 + c][j] += xv * d[j];  // static index: stays in VGPRs
          }
        }
  }
  // combine lanes with equal cq inside the wave (lane = k*CQ + cq)
  for (int o = CQ; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] += __shfl_xor(acc[t][j], o, 64);
  }
  __shared__ float red[4][27][64];  // [wave][t][co]  (Co <= 64)
  if (lane < CQ) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][t][cq * 4 + j] = acc[t][j];
  }
  __syncthreads();
  (void)TT;
  for (int i = threadIdx.x; i < 27 * Co; i += 256) {
    int t = i / Co, co = i - t * Co;  // t = a*9 + b*3 + c slot
    float s = red[0][t][co] + red[1][t][co] + red[2][t][co] + red[3][t][co];
    partial[(((size_t)blockIdx.x * Cin + ci) * 27 + t) * Co + co] = s;
  }
}

__global__ __launch_bounds__(256) void stem_wgrad_finalize(const float* __restrict__ partial, int nch, int Cin, int ky, int kx, int TT, int Co,
                                                           float* __restrict__ dw) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per (co, ci, t) of the torch layout
  if (i >= Co * Cin * TT) return;
  int co = i / (Cin * TT), r = i - co * (Cin * TT), ci = r / TT, t = r - ci * TT;
  int a = t / (ky * kx), b = (t / kx) % ky, c = t % kx;
  int slot = a * 9 + b * 3 + c;
  double s = 0.0;
  for (int k = lane; k < nch; k += 64) s += (double)partial[(((size_t)k * Cin + ci) * 27 + slot) * Co + co];
  s = wave_sum_d(s);
  if (lane == 0) dw[i] = (float)s;
}

int rx_stem_wgrad_mfma_try(rx_dtype dt, const float* x, int n, int cin, int z, int y, int xx, const rx_act* dy, const int32_t kernel[3],
                           float* partial, int max_blocks, int* nblocks_out, hipStream_t st);
#define RX_STEM_CHUNKS 1024
extern "C" size_t rx_stem_conv_bwd_weight_workspace(int cin, int cout, int taps) {
  (void)taps;
  return (size_t)RX_STEM_CHUNKS * cin * 27 * cout * sizeof(float) + 256;
}

extern "C" int rx_stem_conv_bwd_weight(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x, const rx_act* dy, float* dw,
                                       const int32_t kernel[3], void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, dy_ = RxActV(dy), kernel_ = RxI3V(kernel)](void* s) { return rx_stem_conv_bwd_weight(dt, x_ncdhw, n, cin, z, y, x, dy_.p(), dw, kernel_.v, ws, ws_bytes, s); });
  int rc;
  if ((rc = check_vec_channels(dy, dt, "rx_stem_conv_bwd_weight(dy)"))) return rc;
  if ((rc = check_kernel13(kernel, "rx_stem_conv_bwd_weight"))) return rc;
  if (!x_ncdhw || !dw || !ws || cin < 1 || cin > 16) RX_FAIL(RX_EUNSUPPORTED, "rx_stem_conv_bwd_weight: 1 <= Cin <= 16");
  const int Co = dy->c;
  if (Co > 64 || Co % 4 || 64 % (Co / 4)) RX_FAIL(RX_EUNSUPPORTED, "rx_stem_conv_bwd_weight: Cout must be 4,8,16,32 or 64 (got %d)", Co);
  if (dy->n != n || dy->z != z || dy->y != y || dy->x != x) RX_FAIL(RX_EINVAL, "rx_stem_conv_bwd_weight: geometry mismatch");
  if (ws_bytes < rx_stem_conv_bwd_weight_workspace(cin, Co, 27)) RX_FAIL(RX_EWORKSPACE, "rx_stem_conv_bwd_weight: workspace too small");
  hipStream_t st0 = (hipStream_t)stream;
  {
    int nb = 0;
    if (rx_stem_wgrad_mfma_try(dt, x_ncdhw, n, cin, z, y, x, dy, kernel, (float*)ws, RX_STEM_CHUNKS, &nb, st0) == 1) {
      const int TT0 = kernel[0] * kernel[1] * kernel[2];
      int tot0 = Co * cin * TT0;
      hipLaunchKernelGGL(stem_wgrad_finalize, dim3((tot0 + 3) / 4), dim3(256), 0, st0, (const float*)ws, nb, cin, kernel[1], kernel[2],
                         TT0, Co, dw);
      RX_CHECK_LAUNCH("rx_stem_conv_bwd_weight(mfma)");
      return RX_OK;
    }
  }
  const long NV = (long)n * z * y * x;
  const int VPB = 256 / (Co / 4);
  long chunk = (NV + RX_STEM_CHUNKS - 1) / RX_STEM_CHUNKS;
  chunk = (chunk + VPB - 1) / VPB * VPB;
  int nch = (int)((NV + chunk - 1) / chunk);
  const int TT = kernel[0] * kernel[1] * kernel[2];
  hipStream_t st = (hipStream_t)stream;
  RX_DISPATCH_DTYPE(dt, T, {
    hipLaunchKernelGGL((stem_wgrad_kernel<T>), dim3(nch, 1, cin), dim3(256), 0, st, x_ncdhw, cin, z, y, x, (const T*)dy->ptr, dy->ld,
                       rx_act_voxels(dy) * dy->ld, Co, kernel[0], kernel[1], kernel[2], n, (int)chunk, (float*)ws);
    int tot = Co * cin * TT;
    hipLaunchKernelGGL(stem_wgrad_finalize, dim3((tot + 3) / 4), dim3(256), 0, st, (const float*)ws, nch, cin, kernel[1], kernel[2], TT, Co,
                       dw);
  });
  RX_CHECK_LAUNCH("rx_stem_conv_bwd_weight");
  return RX_OK;
}

// ---- weight packing --------------------------------------------------------------------------
// in: w[A][B][T] fp32.  same[t'][A][B], swap[t''][B][A] where t' / t'' optionally reversed.
// One block handles a 32(A) x 32(B) tile for all T taps through LDS.
// Weight packing runs on EVERY training step (all 68 conv / convT weights of cfg2, 1.7 GB of traffic) on the side stream.
// A block owns a 32(A) x 32(B) tile for all TT taps.  The tile is transposed into LDS as [t][a][b] in the compute dtype
// (80-byte b-rows: 16-byte aligned, 16 consecutive rows hit 16 distinct bank slots) while it is loaded with 16-byte
// global reads; `same[t][a][b..b+7]` then leaves as one 16-byte LDS read + one 16-byte store, `swap[t][b][a..a+7]` as
// eight 2-byte LDS reads + one 16-byte store.  (The first version stored every bf16 element with its own 2-byte global
// store and two runtime integer divisions: 97 us per launch, 6.4 ms of kernel time per step.)
#define RX_PACK_PB 40   // LDS pitch of a b-row in elements (80 bytes)
// 1024 threads per tile: the load loop is one 16-byte load + four 2-byte LDS writes per iteration, so the loads in flight per
// CU scale with the thread count (one workgroup per CU for the 512-channel weights: 256 tiles).  256 threads: 34 us per 7 M-parameter
// weight; see DESIGN.md row ai
#ifndef RX_PACK_THREADS
#define RX_PACK_THREADS 1024
#endif

template <typename T>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, int A, int B, int TT, unsigned inv_tt, T* __restrict__ same,
                                          int flip_same, T* __restrict__ swp, int flip_swap, const int a0, const int b0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pack_smem[];
  T* L = reinterpret_cast<T*>(pack_smem);               // [TT][32 a][RX_PACK_PB]
  const int rowlen = 32 * TT;
  const bool full = a0 + 32 <= A && b0 + 32 <= B && sizeof(T) == 2 && (B & 7) == 0 && (A & 7) == 0;
  // ---- load + convert + transpose into LDS
  if (full && TT > 1 && (rowlen & 3) == 0 && ((size_t)B * TT & 3) == 0) {
    const int q4 = rowlen >> 2;                           // float4 pieces per a-row
    for (int q = threadIdx.x; q < 32 * q4; q += RX_PACK_THREADS) {
      const int a = q / q4, c = q - a * q4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((size_t)(a0 + a) * B + b0) * TT + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned r = 4 * c + j;                     // r = b*TT + t
        const unsigned b = __umulhi(r, inv_tt), t = r - b * TT;
        L[((int)t * 32 + a) * RX_PACK_PB + (int)b] = Elem<T>::from_f(v[j]);
      }
    }
  } else {
    for (int i = threadIdx.x; i < 32 * rowlen; i += RX_PACK_THREADS) {
      const int a = i / rowlen, r = i - a * rowlen;
      const int b = r / TT, t = r - b * TT;
      float v = 0.f;
      if (a0 + a < A && b0 + b < B) v = w[((size_t)(a0 + a) * B + b0) * TT + r];
      L[(t * 32 + a) * RX_PACK_PB + b] = Elem<T>::from_f(v);
    }
  }
  __syncthreads();
  if (full) {
    // ---- 16-byte stores: 4 vectors of 8 per (t, row)
    for (int v = threadIdx.x; v < TT * 128; v += RX_PACK_THREADS) {
      const int t = v >> 7, rem = v & 127, row = rem >> 2, c8 = (rem & 3) * 8;
      if (same) {   // row = a, 8 consecutive b
        const int to = flip_same ? TT - 1 - t : t;
        const u32x4 x = *reinterpret_cast<const u32x4*>(L + (t * 32 + row) * RX_PACK_PB + c8);
        *reinterpret_cast<u32x4*>(same + ((size_t)to * A + a0 + row) * B + b0 + c8) = x;
      }
      if (swp) {    // row = b, 8 consecutive a
        const int to = flip_swap ? TT - 1 - t : t;
        T vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[j] = L[(t * 32 + c8 + j) * RX_PACK_PB + row];
        *reinterpret_cast<u32x4*>(swp + ((size_t)to * B + b0 + row) * A + a0 + c8) = *reinterpret_cast<u32x4*>(vals);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < TT * 32 * 32; i += RX_PACK_THREADS) {
    const int t = i / 1024, r = i - t * 1024;
    {
      const int a = r >> 5, b = r & 31;
      if (same && a0 + a < A && b0 + b < B) {
        const int to = flip_same ? TT - 1 - t : t;
        same[((size_t)to * A + a0 + a) * B + b0 + b] = L[(t * 32 + a) * RX_PACK_PB + b];
      }
    }
    {
      const int b = r >> 5, a = r & 31;
      if (swp && a0 + a < A && b0 + b < B) {
        const int to = flip_swap ? TT - 1 - t : t;
        swp[((size_t)to * B + b0 + b) * A + a0 + a] = L[(t * 32 + a) * RX_PACK_PB + b];
      }
    }
  }
}

// more than 27 taps (5- / 7-wide kernels, stride-3 / -4 transposed convs): the [TT][32][40] LDS tile of pack_tile does not fit;
// these layers are rare and small -- one thread per weight, coalesced reads, scattered 2-byte writes
template <typename T>
__global__ __launch_bounds__(256) void pack_naive_kernel(const float* __restrict__ w, int A, int B, int TT, T* __restrict__ same, T* __restrict__ swp) {
  const long total = (long)A * B * TT;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i % TT);
    const long ab = i / TT;
    const int b = (int)(ab % B), a = (int)(ab / B);
    const T v = Elem<T>::from_f(w[i]);
    if (same) same[((long)t * A + a) * B + b] = v;
    if (swp) swp[((long)t * B + b) * A + a] = v;
  }
}
template <typename T>
static void pack_naive_launch(hipStream_t st, const float* w, int A, int B, int TT, void* same, void* swp) {
  long blocks = ((long)A * B * TT + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((pack_naive_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, w, A, B, TT, (T*)same, (T*)swp);
}

template <typename T>
__global__ __launch_bounds__(RX_PACK_THREADS) void pack_kernel(const float* __restrict__ w, int A, int B, int TT, unsigned inv_tt, T* __restrict__ same,
                                                   int flip_same, T* __restrict__ swp, int flip_swap) {
  pack_tile<T>(w, A, B, TT, inv_tt, same, flip_same, swp, flip_swap, blockIdx.y * 32, blockIdx.x * 32);
}

// Table-driven pack: up to RX_PM_MAX weight tensors per launch (pointer / shape table in the kernel arguments, workgroup ->
// tensor by binary search over the first-tile index, as adamw_multi_kernel does).  A cfg2 step re-packs 66 tensors; one
// launch each averaged 17 us (1.13 ms per step on the side stream, 1.5 TB/s: the small tensors are launch-bound).
#define RX_PM_MAX 40
struct PackMulti {
  const float* w[RX_PM_MAX];
  void* same[RX_PM_MAX];
  void* swp[RX_PM_MAX];
  int A[RX_PM_MAX], B[RX_PM_MAX], TT[RX_PM_MAX];
  unsigned inv_tt[RX_PM_MAX];
  int start[RX_PM_MAX + 1];       // first workgroup of tensor i; start[count] = grid size
  int count;
};

template <typename T>
__global__ __launch_bounds__(RX_PACK_THREADS) void pack_multi_kernel(const PackMulti tab) {
  int lo = 0, hi = tab.count - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {               // last i with start[i] <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (tab.start[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  const int i = lo, local = blk - tab.start[i];
  const int nb = (tab.B[i] + 31) >> 5;
  const int ta = local / nb, tb = local - ta * nb;
  pack_tile<T>(tab.w[i], tab.A[i], tab.B[i], tab.TT[i], tab.inv_tt[i], (T*)tab.same[i], 0, (T*)tab.swp[i], 0, ta * 32, tb * 32);
}

// ---- AdamW fused with the weight re-pack (and with gradient clipping) ------------------------------------------------
// The train step ends with clip_grad_norm_ (norm pass + a scale pass over all gradients), the optimizer update (7 fp32
// accesses per parameter) and -- at the start of the next forward -- the re-pack of every conv weight (another read of the
// parameter, two compute-dtype writes): 3.7 ms of kernel time per cfg2 step in 150 launches.  This kernel does the
// last three in ONE pass over a conv / convT weight: the pack kernel's 32 x 32 x T tile walk reads p, g, m, v, applies
// g *= clip (device scalar), the decoupled-weight-decay Adam update (torch.optim.AdamW arithmetic), writes p, m, v back and
// hands the updated tile to the transposed LDS stage of the pack.  `adamw_flat_kernel` is the same update for the
// parameters that are not packed (stem, biases, heads).  Measured: 1 ms less kernel time per step, same wall time (the
// update moves from a side-stream pack that overlapped the forward to the serial end of the step) -> opt-in.
struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;   // bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t)
  float omb1, omb2;                                            // 1 - beta, rounded from double like torch does
};

__device__ inline float adamw_update(float p, float g, float& m, float& v, const AdamArgs a) {
  p -= a.lr * a.weight_decay * p;
  m += a.omb1 * (g - m);                               // lerp(m, g, 1 - beta1)
  v = a.beta2 * v + a.omb2 * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  return p - (a.lr / a.bc1) * (m / denom);
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_pack_kernel(float* __restrict__ w, const float* __restrict__ grad, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ clip, const AdamArgs aa, int A,
                                                         int B, int TT, unsigned inv_tt, T* __restrict__ same, T* __restrict__ swp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pack_smem[];
  T* L = reinterpret_cast<T*>(pack_smem);               // [TT][32 a][RX_PACK_PB]
  const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  const int rowlen = 32 * TT;
  const float cs = clip ? *clip : 1.f;
  const bool full = a0 + 32 <= A && b0 + 32 <= B && sizeof(T) == 2 && (B & 7) == 0 && (A & 7) == 0;
  if (full && TT > 1 && (rowlen & 3) == 0 && ((size_t)B * TT & 3) == 0) {
    const int q4 = rowlen >> 2;
    for (int q = threadIdx.x; q < 32 * q4; q += 256) {
      const int a = q / q4, c = q - a * q4;
      const size_t off = ((size_t)(a0 + a) * B + b0) * TT + 4 * c;
      f32x4 pw = *reinterpret_cast<const f32x4*>(w + off);
      const f32x4 pg = *reinterpret_cast<const f32x4*>(grad + off);
      f32x4 pm = *reinterpret_cast<const f32x4*>(m + off), pv = *reinterpret_cast<const f32x4*>(v + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float mj = pm[j], vj = pv[j];
        pw[j] = adamw_update(pw[j], pg[j] * cs, mj, vj, aa);
        pm[j] = mj, pv[j] = vj;
      }
      *reinterpret_cast<f32x4*>(w + off) = pw;
      *reinterpret_cast<f32x4*>(m + off) = pm;
      *reinterpret_cast<f32x4*>(v + off) = pv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned r = 4 * c + j;
        const unsigned b = __umulhi(r, inv_tt), t = r - b * TT;
        L[((int)t * 32 + a) * RX_PACK_PB + (int)b] = Elem<T>::from_f(pw[j]);
      }
    }
  } else {
    for (int i = threadIdx.x; i < 32 * rowlen; i += 256) {
      const int a = i / rowlen, r = i - a * rowlen;
      const int b = r / TT, t = r - b * TT;
      float nw = 0.f;
      if (a0 + a < A && b0 + b < B) {
        const size_t off = ((size_t)(a0 + a) * B + b0) * TT + r;
        float mj = m[off], vj = v[off];
        nw = adamw_update(w[off], grad[off] * cs, mj, vj, aa);
        w[off] = nw, m[off] = mj, v[off] = vj;
      }
      L[(t * 32 + a) * RX_PACK_PB + b] = Elem<T>::from_f(nw);
    }
  }
  __syncthreads();
  if (full) {
    for (int vv = threadIdx.x; vv < TT * 128; vv += 256) {
      const int t = vv >> 7, rem = vv & 127, row = rem >> 2, c8 = (rem & 3) * 8;
      if (same) {
        const u32x4 x = *reinterpret_cast<const u32x4*>(L + (t * 32 + row) * RX_PACK_PB + c8);
        *reinterpret_cast<u32x4*>(same + ((size_t)t * A + a0 + row) * B + b0 + c8) = x;
      }
      if (swp) {
        T vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[j] = L[(t * 32 + c8 + j) * RX_PACK_PB + row];
        *reinterpret_cast<u32x4*>(swp + ((size_t)t * B + b0 + row) * A + a0 + c8) = *reinterpret_cast<u32x4*>(vals);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < TT * 32 * 32; i += 256) {
    const int t = i / 1024, r = i - t * 1024;
    {
      const int a = r >> 5, b = r & 31;
      if (same && a0 + a < A && b0 + b < B) same[((size_t)t * A + a0 + a) * B + b0 + b] = L[(t * 32 + a) * RX_PACK_PB + b];
    }
    {
      const int b = r >> 5, a = r & 31;
      if (swp && a0 + a < A && b0 + b < B) swp[((size_t)t * B + b0 + b) * A + a0 + a] = L[(t * 32 + a) * RX_PACK_PB + b];
    }
  }
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ w, const float* __restrict__ grad, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ clip, const AdamArgs aa, long n) {
  const float cs = clip ? *clip : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float mj = m[i], vj = v[i];
    w[i] = adamw_update(w[i], grad[i] * cs, mj, vj, aa);
    m[i] = mj, v[i] = vj;
  }
}

static AdamArgs adam_args(double lr, double beta1, double beta2, double eps, double wd, int step) {
  AdamArgs a;
  a.lr = (float)lr, a.beta1 = (float)beta1, a.beta2 = (float)beta2, a.eps = (float)eps, a.weight_decay = (float)wd;
  a.omb1 = (float)(1.0 - beta1), a.omb2 = (float)(1.0 - beta2);
  a.bc1 = (float)(1.0 - pow(beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  return a;
}

// p (A,B,T) fp32 conv weight (kind 0: A = Co, B = Ci -> w_fwd = [t][A][B], w_bwd = [t][B][A]; kind 1: transposed conv,
// A = Ci, B = Co -> w_bwd = [t][A][B], w_fwd = [t][B][A]).  `clip` = optional device scalar multiplied into the gradient.
extern "C" int rx_adamw_pack(rx_dtype dt, float* p, const float* grad, float* exp_avg, float* exp_avg_sq, const float* clip, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int step, int kind, int A, int B, int taps,
                             void* w_fwd, void* w_bwd, void* stream) {
  if (!p || !grad || !exp_avg || !exp_avg_sq || A < 1 || B < 1 || taps < 1 || taps > 27 || step < 1)
    RX_FAIL(RX_EINVAL, "rx_adamw_pack: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const AdamArgs aa = adam_args(lr, beta1, beta2, eps, weight_decay, step);
  void* same = kind == 0 ? w_fwd : w_bwd;
  void* swp = kind == 0 ? w_bwd : w_fwd;
  RX_DISPATCH_DTYPE(dt, T, {
    size_t lds = (size_t)taps * 32 * RX_PACK_PB * sizeof(T);
    const unsigned inv_tt = (unsigned)(((1ull << 32) + taps - 1) / taps);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adamw_pack_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((adamw_pack_kernel<T>), dim3((B + 31) / 32, (A + 31) / 32), dim3(256), lds, st, p, grad, exp_avg, exp_avg_sq, clip, aa,
                       A, B, taps, inv_tt, (T*)same, (T*)swp);
  });
  RX_CHECK_LAUNCH("rx_adamw_pack");
  return RX_OK;
}

extern "C" int rx_adamw_flat(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, const float* clip, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int step, long n, void* stream) {
  if (!p || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) RX_FAIL(RX_EINVAL, "rx_adamw_flat: bad arguments");
  const AdamArgs aa = adam_args(lr, beta1, beta2, eps, weight_decay, step);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, exp_avg, exp_avg_sq, clip, aa, n);
  RX_CHECK_LAUNCH("rx_adamw_flat");
  return RX_OK;
}

// the same update for a LIST of tensors that share hyper-parameters and step count (one optimizer param group): one call from
// the host language instead of one per parameter, and ONE launch per 48 tensors instead of one each.  cfg2 has 69 un-packed
// ... and, with RX_ENGINE_ADAMW=2, every conv weight too: 102 M parameters x 28 B (read p, g, m, v; write p, m, v) = 2.9 GB,
// 0.36 ms at 8 TB/s.  One scalar-load launch per tensor took 1.29 ms per step (18.7 us average over 69 launches: the small
// ones are launch-bound, the large ones ran 4-byte loads); the table kernel below runs 4 x 16-byte loads per array per thread,
// all 16 issued before the first use.  Pointer arrays are HOST arrays.
#define RX_AM_MAX 48
#define RX_AM_CHUNK 4096          // elements per workgroup: 256 threads x 4 float4
struct AdamMulti {
  float* p[RX_AM_MAX];
  const float* g[RX_AM_MAX];
  float* m[RX_AM_MAX];
  float* v[RX_AM_MAX];
  long n[RX_AM_MAX];
  int start[RX_AM_MAX + 1];       // first workgroup of tensor i; start[count] = grid size
  int count;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamMulti t, const float* __restrict__ clip, const AdamArgs aa) {
  const int b = blockIdx.x;
  int lo = 0, hi = t.count;       // start[lo] <= b < start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.start[mid] <= b) lo = mid; else hi = mid;
  }
  const long base = (long)(b - t.start[lo]) * RX_AM_CHUNK;
  const long n = t.n[lo];
  float* __restrict__ w = t.p[lo];
  const float* __restrict__ grad = t.g[lo];
  float* __restrict__ m = t.m[lo];
  float* __restrict__ v = t.v[lo];
  const float cs = clip ? *clip : 1.f;
  const bool vec = (((uintptr_t)w | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec && base + RX_AM_CHUNK <= n) {
    f32x4 pw[4], pg[4], pm[4], pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = base + (long)(j * 256 + threadIdx.x) * 4;
      pw[j] = *reinterpret_cast<const f32x4*>(w + i);
      pg[j] = *reinterpret_cast<const f32x4*>(grad + i);
      pm[j] = *reinterpret_cast<const f32x4*>(m + i);
      pv[j] = *reinterpret_cast<const f32x4*>(v + i);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = base + (long)(j * 256 + threadIdx.x) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float mj = pm[j][k], vj = pv[j][k];
        pw[j][k] = adamw_update(pw[j][k], pg[j][k] * cs, mj, vj, aa);
        pm[j][k] = mj, pv[j][k] = vj;
      }
      *reinterpret_cast<f32x4*>(w + i) = pw[j];
      *reinterpret_cast<f32x4*>(m + i) = pm[j];
      *reinterpret_cast<f32x4*>(v + i) = pv[j];
    }
    return;
  }
  const long end = base + RX_AM_CHUNK < n ? base + RX_AM_CHUNK : n;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float mj = m[i], vj = v[i];
    w[i] = adamw_update(w[i], grad[i] * cs, mj, vj, aa);
    m[i] = mj, v[i] = vj;
  }
}

extern "C" int rx_adamw_flat_multi(int count, float* const* p, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                                   const long* numel, const float* clip, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int step, void* stream) {
  if (count < 0 || (count > 0 && (!p || !grad || !exp_avg || !exp_avg_sq || !numel)) || step < 1) RX_FAIL(RX_EINVAL, "rx_adamw_flat_multi: bad arguments");
  const AdamArgs aa = adam_args(lr, beta1, beta2, eps, weight_decay, step);
  for (int i = 0; i < count; ++i)
    if (!p[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 1) RX_FAIL(RX_EINVAL, "rx_adamw_flat_multi: bad tensor %d", i);
  static const bool per_tensor = [] { const char* e = getenv("RX_ADAMW_PER_TENSOR"); return e && e[0] == '1'; }();   // A/B: the one-launch-per-tensor path
  if (per_tensor) {
    for (int i = 0; i < count; ++i) {
      long blocks = (numel[i] + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p[i], grad[i], exp_avg[i], exp_avg_sq[i], clip,
                         aa, numel[i]);
    }
    RX_CHECK_LAUNCH("rx_adamw_flat_multi");
    return RX_OK;
  }
  for (int i0 = 0; i0 < count;) {
    AdamMulti t;
    int k = 0;
    long blocks = 0;
    for (; i0 + k < count && k < RX_AM_MAX; ++k) {
      const long nb = (numel[i0 + k] + RX_AM_CHUNK - 1) / RX_AM_CHUNK;
      if (blocks + nb > 0x3fffffffL) break;           // keep the grid inside int range
      t.p[k] = p[i0 + k], t.g[k] = grad[i0 + k], t.m[k] = exp_avg[i0 + k], t.v[k] = exp_avg_sq[i0 + k], t.n[k] = numel[i0 + k];
      t.start[k] = (int)blocks;
      blocks += nb;
    }
    if (k == 0) RX_FAIL(RX_EINVAL, "rx_adamw_flat_multi: tensor %d too large", i0);
    for (int q = k; q <= RX_AM_MAX; ++q) t.start[q] = (int)blocks;
    for (int q = k; q < RX_AM_MAX; ++q) t.p[q] = nullptr, t.g[q] = nullptr, t.m[q] = nullptr, t.v[q] = nullptr, t.n[q] = 0;
    t.count = k;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, clip, aa);
    i0 += k;
  }
  RX_CHECK_LAUNCH("rx_adamw_flat_multi");
  return RX_OK;
}

// ---- global gradient norm + clip coefficient (torch.nn.utils.clip_grad_norm_, train.py:227) as two launches ---------------
// torch's path is _foreach_norm (one multi-tensor launch per ~20 tensors: 10 launches of 24 us at cfg2) + stack + vector_norm +
// the scalar arithmetic of the coefficient: ~20 launches at the serial end of a step.  Here: the AdamW table layout (48 tensors
// per launch, one workgroup per 4096 elements) writes one fp32 sum of squares per workgroup, and a single workgroup adds them
// in fp64 in a fixed order (deterministic) and leaves (norm, min(1, max_norm / (norm + 1e-6))) behind.
struct SqnormMulti {
  const float* g[RX_AM_MAX];
  long n[RX_AM_MAX];
  int start[RX_AM_MAX + 1];
  int count;
};

__global__ __launch_bounds__(256) void sqnorm_multi_kernel(const SqnormMulti t, float* __restrict__ partial) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  int lo = 0, hi = t.count;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.start[mid] <= b) lo = mid; else hi = mid;
  }
  const long base = (long)(b - t.start[lo]) * RX_AM_CHUNK;
  const long n = t.n[lo];
  const float* __restrict__ g = t.g[lo];
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0 && base + RX_AM_CHUNK <= n) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(g + base + (long)(j * 256 + threadIdx.x) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) s += v[j][k] * v[j][k];
  } else {
    const long end = base + RX_AM_CHUNK < n ? base + RX_AM_CHUNK : n;
    for (long i = base + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// (one workgroup of 1024 threads, eight loads in flight per thread: the 52 K partials of cfg2 took 64 us as a 256-thread serial
// load chain at the serial end of the step; the summation order stays fixed)
__global__ __launch_bounds__(1024) void sqnorm_finalize_kernel(const float* __restrict__ partial, int nblocks, float max_norm,
                                                               float* __restrict__ out /* [2]: norm, clip coefficient */) {
  __shared__ double red[1024];
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int i = threadIdx.x;
  for (; i + 7 * 1024 < nblocks; i += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += (double)v[u];
  }
  for (; i < nblocks; i += 1024) s[0] += (double)partial[i];
  red[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    const float coef = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = coef < 1.f ? coef : 1.f;
  }
}

// number of fp32 partials rx_grad_norm_clip needs for these tensor sizes
extern "C" long rx_grad_norm_clip_partials(int count, const long* numel) {
  long blocks = 0;
  for (int i = 0; i < count; ++i) blocks += (numel[i] + RX_AM_CHUNK - 1) / RX_AM_CHUNK;
  return blocks;
}

// out[0] = || (g_0, ..., g_{count-1}) ||_2, out[1] = min(1, max_norm / (out[0] + 1e-6)).  `partial`: device scratch of
// rx_grad_norm_clip_partials() floats.  Pointer arrays are HOST arrays.
extern "C" int rx_grad_norm_clip(int count, const float* const* grad, const long* numel, float max_norm, float* partial, long partial_len,
                                 float* out, void* stream) {
  if (count < 1 || !grad || !numel || !partial || !out) RX_FAIL(RX_EINVAL, "rx_grad_norm_clip: bad arguments");
  const long need = rx_grad_norm_clip_partials(count, numel);
  if (need > partial_len || need > 0x3fffffffL) RX_FAIL(RX_EWORKSPACE, "rx_grad_norm_clip: %ld partials needed, %ld given", need, partial_len);
  long done = 0;
  for (int i0 = 0; i0 < count;) {
    SqnormMulti t;
    int k = 0;
    long blocks = 0;
    for (; i0 + k < count && k < RX_AM_MAX; ++k) {
      if (!grad[i0 + k] || numel[i0 + k] < 1) RX_FAIL(RX_EINVAL, "rx_grad_norm_clip: bad tensor %d", i0 + k);
      t.g[k] = grad[i0 + k], t.n[k] = numel[i0 + k];
      t.start[k] = (int)blocks;
      blocks += (numel[i0 + k] + RX_AM_CHUNK - 1) / RX_AM_CHUNK;
    }
    for (int q = k; q <= RX_AM_MAX; ++q) t.start[q] = (int)blocks;
    for (int q = k; q < RX_AM_MAX; ++q) t.g[q] = nullptr, t.n[q] = 0;
    t.count = k;
    hipLaunchKernelGGL(sqnorm_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, partial + done);
    done += blocks;
    i0 += k;
  }
  hipLaunchKernelGGL(sqnorm_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)partial, (int)done, max_norm, out);
  RX_CHECK_LAUNCH("rx_grad_norm_clip");
  return RX_OK;
}

static int pack_generic(rx_dtype dt, const float* w, int A, int B, int TT, void* same, int flip_same, void* swp, int flip_swap,
                        void* stream) {
  if (!w || A < 1 || B < 1 || TT < 1 || TT > RX_MAX_TAPS - 1) RX_FAIL(RX_EINVAL, "rx_pack: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (TT > 27) {
    if (flip_same || flip_swap) RX_FAIL(RX_EUNSUPPORTED, "rx_pack: flipped packs exist for <= 27 taps only");
    RX_DISPATCH_DTYPE(dt, T, pack_naive_launch<T>(st, w, A, B, TT, same, swp));
    RX_CHECK_LAUNCH("rx_pack(naive)");
    return RX_OK;
  }
  RX_DISPATCH_DTYPE(dt, T, {
    size_t lds = (size_t)TT * 32 * RX_PACK_PB * sizeof(T);
    const unsigned inv_tt = (unsigned)(((1ull << 32) + TT - 1) / TT);   // r / TT == umulhi(r, inv_tt) for r < 2^16
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pack_kernel<T>), dim3((B + 31) / 32, (A + 31) / 32), dim3(RX_PACK_THREADS), lds, st, w, A, B, TT, inv_tt, (T*)same, flip_same, (T*)swp,
                       flip_swap);
  });
  RX_CHECK_LAUNCH("rx_pack");
  return RX_OK;
}

extern "C" int rx_pack_conv_weight(rx_dtype dt, const float* w, int co, int ci, int taps, void* w_fwd, void* w_bwd, void* stream) {
  RX_RECORD(stream, [=](void* s) { return rx_pack_conv_weight(dt, w, co, ci, taps, w_fwd, w_bwd, s); });
  // w (Co,Ci,T): w_fwd[t][co][ci] = same; w_bwd[t][ci][co] = swap
  return pack_generic(dt, w, co, ci, taps, w_fwd, 0, w_bwd, 0, stream);
}
extern "C" int rx_pack_convT_weight(rx_dtype dt, const float* w, int ci, int co, int taps, void* w_fwd, void* w_bwd, void* stream) {
  RX_RECORD(stream, [=](void* s) { return rx_pack_convT_weight(dt, w, ci, co, taps, w_fwd, w_bwd, s); });
  // w (Ci,Co,T): w_fwd[t][co][ci] = swap; w_bwd[t][ci][co] = same
  return pack_generic(dt, w, ci, co, taps, w_bwd, 0, w_fwd, 0, stream);
}

// `count` weights in ceil(count / RX_PM_MAX) launches.  kind[i] 0: Conv3d weight (A = Co, B = Ci), 1: ConvTranspose3d weight
// (A = Ci, B = Co); w_fwd[i] / w_bwd[i] as in rx_pack_conv_weight / rx_pack_convT_weight (either may be NULL).  HOST arrays.
extern "C" int rx_pack_multi(rx_dtype dt, int count, const float* const* w, const int* kind, const int* A, const int* B, const int* taps,
                             void* const* w_fwd, void* const* w_bwd, void* stream) {
  if (count < 1 || !w || !kind || !A || !B || !taps || !w_fwd || !w_bwd) RX_FAIL(RX_EINVAL, "rx_pack_multi: bad arguments");
  RxRecScope rx_scope__;
  if (rx_scope__.rec) {       // host arrays: the program keeps its own copies
    std::vector<const float*> w_(w, w + count);
    std::vector<int> kind_(kind, kind + count), A_(A, A + count), B_(B, B + count), taps_(taps, taps + count);
    std::vector<void*> f_(w_fwd, w_fwd + count), b_(w_bwd, w_bwd + count);
    rx_rec_push(RxCmdFn([=](void* s) { return rx_pack_multi(dt, count, w_.data(), kind_.data(), A_.data(), B_.data(), taps_.data(), f_.data(), b_.data(), s); }),
                stream, __func__);
  }
  for (int i = 0; i < count; ++i)
    if (!w[i] || A[i] < 1 || B[i] < 1 || taps[i] < 1 || taps[i] > RX_MAX_TAPS - 1 || (kind[i] != 0 && kind[i] != 1))
      RX_FAIL(RX_EINVAL, "rx_pack_multi: bad entry %d", i);
  hipStream_t st = (hipStream_t)stream;
  {   // entries with more than 27 taps: one naive launch each, the rest goes through the table kernel
    std::vector<const float*> w2;
    std::vector<int> kind2, A2, B2, taps2;
    std::vector<void*> f2, b2;
    bool any_big = false;
    for (int i = 0; i < count; ++i) {
      if (taps[i] > 27) {
        any_big = true;
        void* same = kind[i] == 0 ? w_fwd[i] : w_bwd[i];
        void* swp = kind[i] == 0 ? w_bwd[i] : w_fwd[i];
        RX_DISPATCH_DTYPE(dt, T, pack_naive_launch<T>(st, w[i], A[i], B[i], taps[i], same, swp));
      } else {
        w2.push_back(w[i]), kind2.push_back(kind[i]), A2.push_back(A[i]), B2.push_back(B[i]), taps2.push_back(taps[i]);
        f2.push_back(w_fwd[i]), b2.push_back(w_bwd[i]);
      }
    }
    if (any_big) {
      RX_CHECK_LAUNCH("rx_pack_multi(naive)");
      if (w2.empty()) return RX_OK;
      // (the nested call must not record itself again: rx_scope__ above already pushed this whole call)
      return rx_pack_multi(dt, (int)w2.size(), w2.data(), kind2.data(), A2.data(), B2.data(), taps2.data(), f2.data(), b2.data(), stream);
    }
  }
  for (int i0 = 0; i0 < count; i0 += RX_PM_MAX) {
    PackMulti t;
    memset(&t, 0, sizeof(t));
    const int k = count - i0 < RX_PM_MAX ? count - i0 : RX_PM_MAX;
    long blocks = 0;
    int max_tt = 1;
    for (int j = 0; j < k; ++j) {
      const int i = i0 + j;
      t.w[j] = w[i], t.A[j] = A[i], t.B[j] = B[i], t.TT[j] = taps[i];
      t.inv_tt[j] = (unsigned)(((1ull << 32) + taps[i] - 1) / taps[i]);
      t.same[j] = kind[i] == 0 ? w_fwd[i] : w_bwd[i];
      t.swp[j] = kind[i] == 0 ? w_bwd[i] : w_fwd[i];
      t.start[j] = (int)blocks;
      blocks += (long)((A[i] + 31) / 32) * ((B[i] + 31) / 32);
      if (taps[i] > max_tt) max_tt = taps[i];
    }
    for (int q = k; q <= RX_PM_MAX; ++q) t.start[q] = (int)blocks;
    t.count = k;
    RX_DISPATCH_DTYPE(dt, T, {
      const size_t lds = (size_t)max_tt * 32 * RX_PACK_PB * sizeof(T);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_multi_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((pack_multi_kernel<T>), dim3((unsigned)blocks), dim3(RX_PACK_THREADS), lds, st, t);
    });
  }
  RX_CHECK_LAUNCH("rx_pack_multi");
  return RX_OK;
}
