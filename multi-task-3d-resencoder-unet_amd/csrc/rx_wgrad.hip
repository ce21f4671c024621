// rx_wgrad.hip -- weight gradients of Conv3d / ConvTranspose3d on MFMA.
//
//   dW[tap][r][c] = sum_q  G[q][r] * X[ q*is + tap.d ][c]          (X zero outside its volume)
//
// Conv3d:           G = dy (rows r = Cout), X = x  (cols c = Cin), is = stride, d = t - pad
// ConvTranspose3d:  G = x  (rows r = Cin),  X = dy (cols c = Cout), is = stride, d = t
// -> in both cases the result is torch's own parameter layout [r][c][T] after the reduce kernel.
//
// The contraction index is the VOXEL, which is the slow axis of both channels-last operands, so
// both MFMA operands need a transposed fetch.  Tiles are staged in LDS as [32-channel panel][voxel]
// [32 ch] rows (64 B for 16-bit types) and 16-bit operands are fetched with ds_read_b64_tr_b16
// (4 voxels x 16 channels per 16-lane group, conflict-free on 64-byte rows); fp32 operands use
// plain ds_read_b32 (the 32x32x2 f32 MFMA takes one value per lane).
//
// Work split: grid = (R/BR * C/BC, taps, ksplit).  Inside a block the 4 waves take different
// k-steps of every 64-voxel tile and each computes the whole BR x BC tile; their accumulators are
// combined through LDS, written as an fp32 slab [split][tap][R][C], and a second kernel sums the
// slabs in a fixed order (deterministic, no float atomics) and transposes to [R][C][T].
#include <stdlib.h>

#include "rx_common.h"


struct WgradGeom {
  int Qz, Qy, Qx, Vq, NQ;
  int R, ldg;
  long g_ss;
  int Zx, Yx, Xx, Cc, ldx;
  long x_ss;
  int isz, isy, isx;
  int ntaps, ksplit, q_per_split, tiles_c;
  RxTap taps[RX_MAX_TAPS];
};


typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// operand fetch: `panel` points at a [64 voxels][32 ch] LDS panel; returns the 32x32x(16|8) fragment
// of k-step `ks` (16 voxels for 16-bit types, 8 voxels for fp32)
template <typename T>
__device__ inline u32x4 fetch_frag(const T* panel, int ks, int lane) {
  if constexpr (sizeof(T) == 2) {
    const int g16 = lane >> 4, half = g16 & 1, h = g16 >> 1, l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3;
    const T* a0 = panel + (ks * 16 + 8 * h + q4) * 32 + 16 * half + 4 * p4;
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 32));
    u32x2 lo = __builtin_bit_cast(u32x2, t0), hi = __builtin_bit_cast(u32x2, t1);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  } else {
    const int i = lane & 31, h = lane >> 5;
    const T* a0 = panel + (ks * 8 + 4 * h) * 32 + i;
    f32x4 v = {a0[0], a0[32], a0[64], a0[96]};
    return __builtin_bit_cast(u32x4, v);
  }
}

template <typename T, int BR, int BC>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ gt, const T* __restrict__ xt, float* __restrict__ slab,
                                                    const WgradGeom g) {
  constexpr int P = Elem<T>::PER16;
  constexpr int NR = BR / 32, NC = BC / 32;
  constexpr int GV = 64 * BR / P / 256;  // G vectors per thread per tile
  constexpr int XV = 64 * BC / P / 256;
  constexpr int KSTEPS = sizeof(T) == 2 ? 4 : 8;  // k-steps per 64-voxel tile
  constexpr int TILE_ELEMS = 64 * (BR + BC);      // elements of T per buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sbuf = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / g.tiles_c, tile_c = blockIdx.x - tile_r * g.tiles_c;
  const int r0 = tile_r * BR, c0 = tile_c * BC;
  const RxTap tp = g.taps[blockIdx.y];
  const int q_begin = blockIdx.z * g.q_per_split;
  const int q_end = min(g.NQ, q_begin + g.q_per_split);

  u32x4 gr[GV], xr[XV];
  auto load_tile = [&](int q0) {
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int i = tid + 256 * j, vox = i / (BR / P), cv = i - vox * (BR / P);
      const int q = q0 + vox;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (q < q_end) {
        const int n = q / g.Vq, vq = q - n * g.Vq;
        v = *reinterpret_cast<const u32x4*>(gt + n * g.g_ss + (long)vq * g.ldg + r0 + cv * P);
      }
      gr[j] = v;
    }
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 256 * j, vox = i / (BC / P), cv = i - vox * (BC / P);
      const int q = q0 + vox;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (q < q_end) {
        const int n = q / g.Vq, vq = q - n * g.Vq;
        const int qx = vq % g.Qx, t = vq / g.Qx;
        const int qy = t % g.Qy, qz = t / g.Qy;
        const int z = qz * g.isz + tp.dz, y = qy * g.isy + tp.dy, x = qx * g.isx + tp.dx;
        if ((unsigned)z < (unsigned)g.Zx && (unsigned)y < (unsigned)g.Yx && (unsigned)x < (unsigned)g.Xx)
          v = *reinterpret_cast<const u32x4*>(xt + n * g.x_ss + ((long)(z * g.Yx + y) * g.Xx + x) * g.ldx + c0 + cv * P);
      }
      xr[j] = v;
    }
  };
  auto store_tile = [&](int buf) {
    T* base = sbuf + buf * TILE_ELEMS;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int i = tid + 256 * j, vox = i / (BR / P), cv = i - vox * (BR / P);
      const int c = cv * P;
      *reinterpret_cast<u32x4*>(base + ((c >> 5) * 64 + vox) * 32 + (c & 31)) = gr[j];
    }
    T* xb = base + 64 * BR;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 256 * j, vox = i / (BC / P), cv = i - vox * (BC / P);
      const int c = cv * P;
      *reinterpret_cast<u32x4*>(xb + ((c >> 5) * 64 + vox) * 32 + (c & 31)) = xr[j];
    }
  };

  f32x16 acc[NR][NC];
#pragma unroll
  for (int a = 0; a < NR; ++a)
#pragma unroll
    for (int b = 0; b < NC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (q_begin < q_end) {
    load_tile(q_begin);
    store_tile(0);
  }
  __syncthreads();
  int it = 0;
  for (int q0 = q_begin; q0 < q_end; q0 += 64, ++it) {
    const int buf = it & 1;
    if (q0 + 64 < q_end) load_tile(q0 + 64);
    const T* base = sbuf + buf * TILE_ELEMS;
#pragma unroll
    for (int kk = 0; kk < KSTEPS / 4; ++kk) {
      const int ks = wave + 4 * kk;
      u32x4 af[NR], bf[NC];
#pragma unroll
      for (int a = 0; a < NR; ++a) af[a] = fetch_frag<T>(base + a * 64 * 32, ks, lane);
#pragma unroll
      for (int b = 0; b < NC; ++b) bf[b] = fetch_frag<T>(base + 64 * BR + b * 64 * 32, ks, lane);
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) Mma<T>::run(acc[a][b], af[a], bf[b]);
    }
    if (q0 + 64 < q_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- combine the 4 waves through LDS (aliases the staging buffers; all reads of them are done)
  float* red = reinterpret_cast<float*>(smem);  // [wave][a][b][row 32][col 32]
  const int col = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int a = 0; a < NR; ++a)
#pragma unroll
    for (int b = 0; b < NC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
        red[(((wave * NR + a) * NC + b) * 32 + row) * 32 + col] = acc[a][b][r];
      }
  __syncthreads();
  float* out = slab + ((long)blockIdx.z * g.ntaps + blockIdx.y) * g.R * g.Cc;
  for (int i = tid; i < NR * NC * 1024; i += 256) {
    const int blk = i >> 10, rc = i & 1023, row = rc >> 5, cc = rc & 31;
    const int a = blk / NC, b = blk - a * NC;
    float s = red[i] + red[i + NR * NC * 1024] + red[i + 2 * NR * NC * 1024] + red[i + 3 * NR * NC * 1024];
    out[(long)(r0 + a * 32 + row) * g.Cc + c0 + b * 32 + cc] = s;
  }
}

// dw[r][c][t] = sum_s slab[s][t][r][c].  A block owns 256 consecutive (r,c) pairs: reads are coalesced over
// (r,c) for every (s,t), the [256][T] result is transposed through LDS and written as one contiguous run.
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ slab, int S, int T_, int R, int C, float* __restrict__ dw) {
  __shared__ float tile[27 * 257];
  const long RC = (long)R * C;
  const long base = (long)blockIdx.x * 256;
  const long i = base + threadIdx.x;
  for (int t = 0; t < T_; ++t) {
    float s = 0.f;
    if (i < RC)
      for (int k = 0; k < S; ++k) s += slab[((long)k * T_ + t) * RC + i];
    tile[t * 257 + threadIdx.x] = s;
  }
  __syncthreads();
  const long n_here = (RC - base < 256 ? RC - base : 256) * T_;
  for (long j = threadIdx.x; j < n_here; j += 256) {
    int pair = (int)(j / T_), t = (int)(j - (long)pair * T_);
    dw[base * T_ + j] = tile[t * 257 + pair];
  }
}

// many-splits variant (small R*C, e.g. 32x32 panels with 512 slabs): one block per (256 pairs, tap); the loop over
// the splits is 8-way unrolled so every wave keeps 8 independent 1-KB loads in flight.  Output writes are
// stride-T but the tensors of this regime are tiny.
__global__ __launch_bounds__(256) void wgrad_reduce_manysplits(const float* __restrict__ slab, int S, int T_, int R, int C,
                                                               float* __restrict__ dw) {
  const long RC = (long)R * C;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  if (i >= RC) return;
  const float* p = slab + (long)t * RC + i;
  const long stride = (long)T_ * RC;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
  int k = 0;
  for (; k + 8 <= S; k += 8) {
    a0 += p[(k + 0) * stride];
    a1 += p[(k + 1) * stride];
    a2 += p[(k + 2) * stride];
    a3 += p[(k + 3) * stride];
    a4 += p[(k + 4) * stride];
    a5 += p[(k + 5) * stride];
    a6 += p[(k + 6) * stride];
    a7 += p[(k + 7) * stride];
  }
  for (; k < S; ++k) a0 += p[k * stride];
  dw[i * T_ + t] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

// ---------------------------------------------------------------------------------------------
static int plan_split(const WgradGeom& g, int BR, int BC, size_t ws_bytes, int* ksplit, int* qps) {
  const long tiles = (long)(g.R / BR) * (g.Cc / BC) * g.ntaps;
  const long ktiles = ((long)g.NQ + 63) / 64;
  long S = (1024 + tiles - 1) / tiles;
  if (S > ktiles / 4) S = ktiles / 4;
  if (S < 1) S = 1;
  const size_t slab1 = (size_t)g.ntaps * g.R * g.Cc * sizeof(float);
  while (S > 1 && S * slab1 > ws_bytes) --S;
  if (slab1 > ws_bytes) return RX_EWORKSPACE;
  long per = (ktiles + S - 1) / S;
  S = (ktiles + per - 1) / per;
  *ksplit = (int)S;
  *qps = (int)(per * 64);
  return RX_OK;
}

static size_t wgrad_ws_bytes(int R, int C, int taps, long NQ) {
  const size_t slab1 = (size_t)taps * R * C * sizeof(float);
  const long tiles = (long)((R + 63) / 64) * ((C + 63) / 64) * taps;
  long S = (1024 + tiles - 1) / tiles;
  long ktiles = (NQ + 63) / 64;
  if (S > ktiles / 4) S = ktiles / 4;
  if (S < 1) S = 1;
  // BR/BC may fall back to 32 (4x the tiles) -> S only shrinks; this bound is safe
  if (R % 64 || C % 64) {
    const long t32 = (long)(R / 32) * (C / 32) * taps;
    long S2 = (1024 + t32 - 1) / t32;
    if (S2 > S) S = S2;
  }
  return S * slab1 + 256;
}

template <typename T>
static void wgrad_dispatch(int BR, int BC, dim3 grid, hipStream_t st, const void* gt, const void* xt, float* slab, const WgradGeom& g) {
  const size_t lds = 65536;
#define RX_WG(BR_, BC_)                                                                                                            \
  do {                                                                                                                             \
    static bool attr_set = false;                                                                                                  \
    if (!attr_set) {                                                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BR_, BC_>),                                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                             \
      attr_set = true;                                                                                                             \
    }                                                                                                                              \
    hipLaunchKernelGGL((wgrad_kernel<T, BR_, BC_>), grid, dim3(256), lds, st, (const T*)gt, (const T*)xt, slab, g);                \
  } while (0)
  if (BR == 64 && BC == 64)
    RX_WG(64, 64);
  else if (BR == 64 && BC == 32)
    RX_WG(64, 32);
  else if (BR == 32 && BC == 64)
    RX_WG(32, 64);
  else
    RX_WG(32, 32);
#undef RX_WG
}

void rx_wgrad_reduce_launch(const float* slab, int S, int T_, int R, int C, float* dw, hipStream_t st);

static int wgrad_launch(rx_dtype dt, const void* gt, const void* xt, float* dw, WgradGeom& g, void* ws, size_t ws_bytes, hipStream_t st) {
  const int per16 = dt == RX_F32 ? 4 : 8;
  if (g.R % 32 || g.Cc % 32) RX_FAIL(RX_EUNSUPPORTED, "wgrad: channel counts must be multiples of 32 (R=%d C=%d)", g.R, g.Cc);
  if (g.ldg % per16 || g.ldx % per16 || ((uintptr_t)gt & 15) || ((uintptr_t)xt & 15)) RX_FAIL(RX_EUNSUPPORTED, "wgrad: misaligned operand");
  if (!ws || !dw) RX_FAIL(RX_EINVAL, "wgrad: null workspace / output");
  const int BR = g.R % 64 == 0 ? 64 : 32, BC = g.Cc % 64 == 0 ? 64 : 32;
  g.tiles_c = g.Cc / BC;
  int rc = plan_split(g, BR, BC, ws_bytes, &g.ksplit, &g.q_per_split);
  if (rc) RX_FAIL(rc, "wgrad: workspace too small (%zu bytes)", ws_bytes);
  dim3 grid((g.R / BR) * (g.Cc / BC), g.ntaps, g.ksplit);
  rx_note_kernel(BR == 64 ? (BC == 64 ? "wgrad_kernel<64,64>" : "wgrad_kernel<64,32>") : (BC == 64 ? "wgrad_kernel<32,64>" : "wgrad_kernel<32,32>"));
  RX_DISPATCH_DTYPE(dt, T, wgrad_dispatch<T>(BR, BC, grid, st, gt, xt, (float*)ws, g));
  rx_wgrad_reduce_launch((const float*)ws, g.ksplit, g.ntaps, g.R, g.Cc, dw, st);
  RX_CHECK_LAUNCH("wgrad");
  return RX_OK;
}

static int conv_out_dim(int in, int k, int s) { return (in + 2 * ((k - 1) / 2) - k) / s + 1; }

// shared with rx_wgrad_halo.hip
void rx_wgrad_reduce_launch(const float* slab, int S, int T_, int R, int C, float* dw, hipStream_t st) {
  long RC = (long)R * C;
  if (S >= 8 || T_ > 27)       // (wgrad_reduce transposes a [256][T <= 27] tile through LDS)
    hipLaunchKernelGGL(wgrad_reduce_manysplits, dim3((unsigned)((RC + 255) / 256), T_), dim3(256), 0, st, slab, S, T_, R, C, dw);
  else
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)((RC + 255) / 256)), dim3(256), 0, st, slab, S, T_, R, C, dw);
}
size_t rx_wgrad_halo_ws_bytes(const rx_act* x, const rx_act* dy);
int rx_wgrad_halo_try(rx_dtype dt, const rx_act* x, const rx_act* dy, const int32_t stride[3], float* dw, void* ws, size_t ws_bytes,
                      hipStream_t st);

extern "C" size_t rx_conv3d_bwd_weight_workspace(const rx_act* x, const rx_act* dy, const int32_t kernel[3]) {
  if (!rx_act_ok_planar(x) || !rx_act_ok(dy)) return 0;
  size_t a = wgrad_ws_bytes(dy->c, x->c, kernel[0] * kernel[1] * kernel[2], (long)dy->n * rx_act_voxels(dy));
  size_t b = (kernel[0] == 3 && kernel[1] == 3 && kernel[2] == 3) ? rx_wgrad_halo_ws_bytes(x, dy) : 0;
  return a > b ? a : b;
}

extern "C" int rx_conv3d_bwd_weight(rx_dtype dt, const rx_act* x, const rx_act* dy, float* dw, const int32_t kernel[3],
                                    const int32_t stride[3], void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), dy_ = RxActV(dy), kernel_ = RxI3V(kernel), stride_ = RxI3V(stride)](void* s) { return rx_conv3d_bwd_weight(dt, x_.p(), dy_.p(), dw, kernel_.v, stride_.v, ws, ws_bytes, s); });
  if (!rx_act_ok_planar(x) || !rx_act_ok(dy)) RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_weight: bad arguments");
  for (int i = 0; i < 3; ++i) {
    if (kernel[i] < 1 || kernel[i] > RX_MAX_KERNEL) RX_FAIL(RX_EUNSUPPORTED, "rx_conv3d_bwd_weight: kernel sizes must be 1..%d per axis", RX_MAX_KERNEL);
    if (stride[i] < 1 || stride[i] > RX_MAX_STRIDE) RX_FAIL(RX_EUNSUPPORTED, "rx_conv3d_bwd_weight: strides must be 1..%d per axis", RX_MAX_STRIDE);
  }
  if (dy->n != x->n || dy->z != conv_out_dim(x->z, kernel[0], stride[0]) || dy->y != conv_out_dim(x->y, kernel[1], stride[1]) ||
      dy->x != conv_out_dim(x->x, kernel[2], stride[2]))
    RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_weight: geometry mismatch");
  if (kernel[0] == 3 && kernel[1] == 3 && kernel[2] == 3 && ws && dw && !(getenv("RX_NO_STRIDED_WGH") && (stride[0] > 1 || stride[1] > 1 || stride[2] > 1))) {
    int rc = rx_wgrad_halo_try(dt, x, dy, stride, dw, ws, ws_bytes, (hipStream_t)stream);  // LDS-halo kernel (16-bit types), stride 1 or 2
    if (rc < 0) return rc;
    if (rc == 1) return RX_OK;
  }
  if (x->cs) RX_FAIL(RX_EUNSUPPORTED, "rx_conv3d_bwd_weight: a planar-concat x needs the LDS-halo weight-gradient kernels");
  WgradGeom g;
  memset(&g, 0, sizeof(g));
  g.Qz = dy->z, g.Qy = dy->y, g.Qx = dy->x, g.Vq = (int)rx_act_voxels(dy), g.NQ = dy->n * g.Vq;
  g.R = dy->c, g.ldg = dy->ld, g.g_ss = (long)g.Vq * dy->ld;
  g.Zx = x->z, g.Yx = x->y, g.Xx = x->x, g.Cc = x->c, g.ldx = x->ld, g.x_ss = rx_act_voxels(x) * (long)x->ld;
  g.isz = stride[0], g.isy = stride[1], g.isx = stride[2];
  const int pz = (kernel[0] - 1) / 2, py = (kernel[1] - 1) / 2, px = (kernel[2] - 1) / 2;
  for (int a = 0; a < kernel[0]; ++a)
    for (int b = 0; b < kernel[1]; ++b)
      for (int c = 0; c < kernel[2]; ++c) {
        RxTap& t = g.taps[g.ntaps];
        t.dz = (int8_t)(a - pz), t.dy = (int8_t)(b - py), t.dx = (int8_t)(c - px);
        t.w = (uint16_t)g.ntaps;
        ++g.ntaps;
      }
  return wgrad_launch(dt, dy->ptr, x->ptr, dw, g, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of nn.ConvTranspose3d(kernel == stride) (decoder.py:110-113):  dW[ci][co][t] = sum_v x[v][ci] dy[s*v + t][co].
// wgrad_kernel runs one workgroup per (tap, split): the coarse tensor x is re-read for every tap (64 -> 32 at 64^3 -> 128^3:
// 587 MB for a 335 MB problem, 222 us at 9.7 TFLOP/s).  Here a workgroup of 8 waves walks a contiguous range of 64-voxel
// coarse tiles; per tile the x panel and the `taps` fine dy panels are staged ONCE (register-prefetched, [32 ch][64 voxels]
// LDS panels read with ds_read_b64_tr_b16 as in wgrad_kernel) and wave w accumulates tap w: every operand byte is read from
// HBM once.  Slabs [split][tap][Ci][Co] + the fixed-order reduce.  16-bit types, taps <= 8, (Ci/32)*(Co/32) <= 8.
// ---------------------------------------------------------------------------------------------------------------------
struct ConvTWgGeom {
  int Vq, NQ, Qy, Qx;         // coarse voxels per sample / in total, coarse Y, X
  int R, ldr;                 // coarse channels (rows of dW)
  long r_ss;
  int Yf, Xf, Cc, ldc;        // fine tensor
  long c_ss;
  int sz, sy, sx, ntaps;
  int q_per_split;
};

template <typename T, int NR, int NC>
__global__ __launch_bounds__(512) void convT_wgrad_kernel(const T* __restrict__ xt, const T* __restrict__ yt, float* __restrict__ slab,
                                                          const ConvTWgGeom g) {
  constexpr int P = Elem<T>::PER16;              // 8
  constexpr int BR = 32 * NR, BC = 32 * NC;
  constexpr int XV = (64 * BR / P + 511) / 512;   // x pieces per thread per tile
  constexpr int YV = (64 * BC / P + 511) / 512;   // dy pieces per thread per tile and tap
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sX = reinterpret_cast<T*>(smem);             // [NR][64][32]
  T* sY = sX + 64 * BR;                           // [tap][NC][64][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q_begin = blockIdx.x * g.q_per_split;
  const int q_end = min(g.NQ, q_begin + g.q_per_split);

  u32x4 xr[XV], yr[8][YV];
  auto load_tile = [&](int q0) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 512 * j, vox = i / (BR / P), cv = i - vox * (BR / P);
      const int q = q0 + vox;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (vox < 64 && q < q_end) {
        const int n = q / g.Vq, vq = q - n * g.Vq;
        v = *reinterpret_cast<const u32x4*>(xt + n * g.r_ss + (long)vq * g.ldr + cv * P);
      }
      xr[j] = v;
    }
#pragma unroll
    for (int j = 0; j < YV; ++j) {
      const int i = tid + 512 * j, vox = i / (BC / P), cv = i - vox * (BC / P);
      const int q = q0 + vox;
      const bool ok = vox < 64 && q < q_end;
      const int n = ok ? q / g.Vq : 0, vq = ok ? q - n * g.Vq : 0;
      const int qx = vq % g.Qx, t2 = vq / g.Qx;
      const int qy = t2 % g.Qy, qz = t2 / g.Qy;
      const T* base = yt + n * g.c_ss + cv * P;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (ok && t < g.ntaps) {
          const int c = t % g.sx, b = (t / g.sx) % g.sy, a = t / (g.sx * g.sy);
          const long fv = ((long)(qz * g.sz + a) * g.Yf + (qy * g.sy + b)) * g.Xf + (qx * g.sx + c);
          v = *reinterpret_cast<const u32x4*>(base + fv * g.ldc);
        }
        yr[t][j] = v;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 512 * j, vox = i / (BR / P), cv = i - vox * (BR / P);
      const int c = cv * P;
      if (vox < 64) *reinterpret_cast<u32x4*>(sX + ((c >> 5) * 64 + vox) * 32 + (c & 31)) = xr[j];
    }
#pragma unroll
    for (int j = 0; j < YV; ++j) {
      const int i = tid + 512 * j, vox = i / (BC / P), cv = i - vox * (BC / P);
      const int c = cv * P;
      if (vox < 64) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < g.ntaps) *reinterpret_cast<u32x4*>(sY + (long)t * 64 * BC + ((c >> 5) * 64 + vox) * 32 + (c & 31)) = yr[t][j];
      }
    }
  };

  f32x16 acc[NR][NC];
#pragma unroll
  for (int a = 0; a < NR; ++a)
#pragma unroll
    for (int b = 0; b < NC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (q_begin < q_end) load_tile(q_begin);
  for (int q0 = q_begin; q0 < q_end; q0 += 64) {
    __syncthreads();                      // everybody is done reading the previous tile
    store_tile();
    __syncthreads();
    if (q0 + 64 < q_end) load_tile(q0 + 64);        // in flight under the MFMAs
    if (wave < g.ntaps) {
      const T* yb = sY + (long)wave * 64 * BC;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 af[NR], bf[NC];
#pragma unroll
        for (int a = 0; a < NR; ++a) af[a] = fetch_frag<T>(sX + a * 64 * 32, ks, lane);
#pragma unroll
        for (int b = 0; b < NC; ++b) bf[b] = fetch_frag<T>(yb + b * 64 * 32, ks, lane);
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
          for (int b = 0; b < NC; ++b) Mma<T>::run(acc[a][b], af[a], bf[b]);
      }
    }
  }
  if (wave < g.ntaps) {
    const int col = lane & 31, fh = lane >> 5;
    float* out = slab + ((long)blockIdx.x * g.ntaps + wave) * g.R * g.Cc;
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
      for (int b = 0; b < NC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
          out[(long)(a * 32 + row) * g.Cc + b * 32 + col] = acc[a][b][r];
        }
  }
}

static int convT_wgrad_splits(long NQ, size_t slab1) {
  long ktiles = (NQ + 63) / 64;
  long S = ktiles / 8;                    // >= 8 tiles per workgroup
  if (S > 256) S = 256;
  while (S > 32 && (size_t)S * slab1 > ((size_t)32 << 20)) S /= 2;      // slabs stay within ~32 MB (MALL-resident round trip)
  if (S < 1) S = 1;
  return (int)S;
}

// 1 = handled, 0 = not applicable (fall through to wgrad_kernel), < 0 error
static int convT_wgrad_try(rx_dtype dt, const rx_act* x, const rx_act* dy, const int32_t stride[3], float* dw, void* ws, size_t ws_bytes,
                           hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("RX_CONVT_WGRAD");
    on = e ? atoi(e) : 1;
  }
  const int taps = stride[0] * stride[1] * stride[2];
  const int R = x->c, C = dy->c;
  if (!on || dt == RX_F32 || taps > 8 || R % 32 || C % 32 || x->ld % 8 || dy->ld % 8 || ((uintptr_t)x->ptr & 15) || ((uintptr_t)dy->ptr & 15)) return 0;
  const int NR = R / 32, NC = C / 32;
  if (!((NR == 2 && NC == 1) || (NR == 4 && NC == 2) || (NR == 1 && NC == 1) || (NR == 2 && NC == 2))) return 0;
  const long NQ = (long)x->n * rx_act_voxels(x);
  if (NQ < 32768) return 0;               // the low-resolution layers: too few tiles to split, the generic kernel is fine there
  const size_t slab1 = (size_t)taps * R * C * sizeof(float);
  const int S = convT_wgrad_splits(NQ, slab1);
  if ((size_t)S * slab1 > ws_bytes) return 0;
  ConvTWgGeom g;
  g.Vq = (int)rx_act_voxels(x), g.NQ = (int)NQ, g.Qy = x->y, g.Qx = x->x;
  g.R = R, g.ldr = x->ld, g.r_ss = (long)g.Vq * x->ld;
  g.Yf = dy->y, g.Xf = dy->x, g.Cc = C, g.ldc = dy->ld, g.c_ss = rx_act_voxels(dy) * (long)dy->ld;
  g.sz = stride[0], g.sy = stride[1], g.sx = stride[2], g.ntaps = taps;
  const long ktiles = (NQ + 63) / 64;
  g.q_per_split = (int)(((ktiles + S - 1) / S) * 64);
  const int S2 = (int)((NQ + g.q_per_split - 1) / g.q_per_split);
  const size_t lds = (size_t)64 * (R + taps * C) * 2;
  rx_note_kernel("convT_wgrad_kernel");
#define RX_CTW(NR_, NC_)                                                                                                              \
  do {                                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_wgrad_kernel<T, NR_, NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((convT_wgrad_kernel<T, NR_, NC_>), dim3(S2), dim3(512), lds, st, (const T*)x->ptr, (const T*)dy->ptr, (float*)ws, g);            \
  } while (0)
#define RX_CTW_ALL()                       \
  if (NR == 2 && NC == 1) RX_CTW(2, 1);    \
  else if (NR == 4 && NC == 2) RX_CTW(4, 2); \
  else if (NR == 1 && NC == 1) RX_CTW(1, 1); \
  else RX_CTW(2, 2)
  if (dt == RX_BF16) {
    using T = bf16_t;
    RX_CTW_ALL();
  } else {
    using T = f16_t;
    RX_CTW_ALL();
  }
#undef RX_CTW_ALL
#undef RX_CTW
  rx_wgrad_reduce_launch((const float*)ws, S2, taps, R, C, dw, st);
  return 1;
}

extern "C" size_t rx_convT3d_bwd_weight_workspace(const rx_act* x, const rx_act* dy, const int32_t stride[3]) {
  if (!rx_act_ok(x) || !rx_act_ok(dy)) return 0;
  const int taps = stride[0] * stride[1] * stride[2];
  const long NQ = (long)x->n * rx_act_voxels(x);
  const size_t slab1 = (size_t)taps * x->c * dy->c * sizeof(float);
  const size_t a = wgrad_ws_bytes(x->c, dy->c, taps, NQ);
  const size_t b = (size_t)convT_wgrad_splits(NQ, slab1) * slab1 + 256;
  return a > b ? a : b;
}

extern "C" int rx_convT3d_bwd_weight(rx_dtype dt, const rx_act* x, const rx_act* dy, float* dw, const int32_t stride[3], void* ws,
                                     size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), dy_ = RxActV(dy), stride_ = RxI3V(stride)](void* s) { return rx_convT3d_bwd_weight(dt, x_.p(), dy_.p(), dw, stride_.v, ws, ws_bytes, s); });
  if (!rx_act_ok(x) || !rx_act_ok(dy)) RX_FAIL(RX_EINVAL, "rx_convT3d_bwd_weight: bad arguments");
  for (int i = 0; i < 3; ++i)
    if (stride[i] < 1 || stride[i] > RX_MAX_STRIDE) RX_FAIL(RX_EUNSUPPORTED, "rx_convT3d_bwd_weight: strides must be 1..%d per axis", RX_MAX_STRIDE);
  if (dy->n != x->n || dy->z != x->z * stride[0] || dy->y != x->y * stride[1] || dy->x != x->x * stride[2])
    RX_FAIL(RX_EINVAL, "rx_convT3d_bwd_weight: geometry mismatch");
  if (ws && dw) {
    const int rc = convT_wgrad_try(dt, x, dy, stride, dw, ws, ws_bytes, (hipStream_t)stream);    // every operand byte once
    if (rc < 0) return rc;
    if (rc == 1) {
      RX_CHECK_LAUNCH("rx_convT3d_bwd_weight(convT_wgrad)");
      return RX_OK;
    }
  }
  // dW[ci][co][t] = sum_i x[i][ci] * dy[i*s + t][co]
  WgradGeom g;
  memset(&g, 0, sizeof(g));
  g.Qz = x->z, g.Qy = x->y, g.Qx = x->x, g.Vq = (int)rx_act_voxels(x), g.NQ = x->n * g.Vq;
  g.R = x->c, g.ldg = x->ld, g.g_ss = (long)g.Vq * x->ld;
  g.Zx = dy->z, g.Yx = dy->y, g.Xx = dy->x, g.Cc = dy->c, g.ldx = dy->ld, g.x_ss = rx_act_voxels(dy) * (long)dy->ld;
  g.isz = stride[0], g.isy = stride[1], g.isx = stride[2];
  for (int a = 0; a < stride[0]; ++a)
    for (int b = 0; b < stride[1]; ++b)
      for (int c = 0; c < stride[2]; ++c) {
        RxTap& t = g.taps[g.ntaps];
        t.dz = (int8_t)a, t.dy = (int8_t)b, t.dx = (int8_t)c;
        t.w = (uint16_t)g.ntaps;
        ++g.ntaps;
      }
  return wgrad_launch(dt, x->ptr, dy->ptr, dw, g, ws, ws_bytes, (hipStream_t)stream);
}
