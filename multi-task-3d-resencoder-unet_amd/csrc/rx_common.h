// rx_common.h -- shared device/host helpers for librxunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rxunet.h"

// Host-side per-launch overhead: the dispatch code asks the environment for tuning knobs and (re)sets the dynamic-LDS limit of
// the kernel it is about to launch on EVERY call (~700 launches per train step).  Both are answered from per-thread caches
// after the first time (rx_prog.hip); the knobs are therefore read once per process, which is how they are used.
const char* rx_getenv_cached(const char* name);
hipError_t rx_func_attr_once(const void* fn, hipFuncAttribute attr, int value);
#ifndef RX_NO_HOST_MACROS
#define getenv(name) rx_getenv_cached(name)
#define hipFuncSetAttribute(fn, attr, value) rx_func_attr_once((fn), (attr), (value))
#endif

#define RX_WAVE 64

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#include "rx_prog.h"

// ---- error plumbing (host) -----------------------------------------------------------------
void rx_set_error(const char* fmt, ...);
void rx_note_kernel(const char* name);   // records which kernel instantiation the last conv entry point used
#define RX_FAIL(code, ...)      \
  do {                          \
    rx_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)
#define RX_CHECK_LAUNCH(name)                                                  \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) RX_FAIL(RX_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// what a backward-data launch needs to know about the InstanceNorm layer (without residual) whose output gradient it completes
// Epilogue stores of the 32x32 MFMA accumulator layout: lane l < 32 and lane l + 32 hold 8-byte pieces of the SAME voxel at
// channels 8*g4 + 4*fh.  One v_permlane32_swap per dword (piece g4 of the upper half <-> piece g4+1 of the lower half) turns
// two 8-byte pieces per lane into ONE 16-byte piece: the lower lane gets channels [8*g4, 8*g4+8), the upper lane
// [8*g4+8, 8*g4+16) -- half as many store instructions, 32 contiguous bytes per voxel and instruction instead of 16.
#ifndef RX_ST16
#define RX_ST16 1
#endif
__device__ inline u32x4 rx_pair16(u32x2 a, u32x2 b) {
  auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
  return u32x4{r0[0], r1[0], r0[1], r1[1]};
}

struct RxBwdStat {
  const rx_act* y;      // saved conv output of that layer
  const float* stats;   // its (mean, rstd)
  float slope;
};

// Ablation switches of the halo kernels (RX_DBG env -> geometry.dbg): compiled in only with -DRX_ABLATION=1.  Run-time flags
// are not free in these kernels (DESIGN §8 row y), so the production build folds them away.
#ifndef RX_ABLATION
#define RX_ABLATION 0
#endif
#define RX_ABLATE(g, bit) (RX_ABLATION && ((g).dbg & (bit)))

// ---- element traits ------------------------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int PER16 = 4;  // elements per 16-byte vector
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
  static constexpr int PER16 = 8;
  __device__ static inline float to_f(bf16_t v) { return (float)v; }
  __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }
};
template <>
struct Elem<f16_t> {
  static constexpr int PER16 = 8;
  __device__ static inline float to_f(f16_t v) { return (float)v; }
  __device__ static inline f16_t from_f(float v) { return (f16_t)v; }
};

// 16-byte vector of T, convertible to/from floats
template <typename T>
struct alignas(16) Vec16 {
  T v[Elem<T>::PER16];
};

template <typename T>
__device__ inline Vec16<T> ld16(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<u32x4*>(&r) = *reinterpret_cast<const u32x4*>(p);
  return r;
}
template <typename T>
__device__ inline void st16(T* p, const Vec16<T>& r) {
  *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(&r);
}
template <typename T>
__device__ inline Vec16<T> zero16() {
  Vec16<T> r;
  *reinterpret_cast<u32x4*>(&r) = u32x4{0u, 0u, 0u, 0u};
  return r;
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- dtype dispatch ------------------------------------------------------------------------
#define RX_DISPATCH_DTYPE(dt, T, ...)                          \
  switch (dt) {                                                \
    case RX_F32: {                                             \
      using T = float;                                         \
      __VA_ARGS__;                                             \
    } break;                                                   \
    case RX_BF16: {                                            \
      using T = bf16_t;                                        \
      __VA_ARGS__;                                             \
    } break;                                                   \
    case RX_F16: {                                             \
      using T = f16_t;                                         \
      __VA_ARGS__;                                             \
    } break;                                                   \
    default:                                                   \
      RX_FAIL(RX_EINVAL, "unknown dtype %d", (int)(dt));       \
  }

static inline size_t rx_dtype_size(int dt) { return dt == RX_F32 ? 4 : 2; }
static inline size_t rx_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline long rx_act_voxels(const rx_act* a) { return (long)a->z * a->y * a->x; }
static inline bool rx_act_ok(const rx_act* a) {
  return a && a->ptr && a->n > 0 && a->z > 0 && a->y > 0 && a->x > 0 && a->c > 0 && a->ld >= a->c && a->cs == 0;
}
// the same, or a planar concat of 32-channel groups (rx_act.cs): only the 3x3x3 stride-1 conv entry points accept it
static inline bool rx_act_ok_planar(const rx_act* a) {
  if (a && a->cs != 0)
    return a->ptr && a->n > 0 && a->z > 0 && a->y > 0 && a->x > 0 && a->ld == 32 && a->c > 32 && a->c % 32 == 0 && a->cs > 0 && a->cs % 8 == 0;
  return rx_act_ok(a);
}

// Epilogues: `bias != nullptr` and `g.accumulate` are launch-uniform, but tested per VALUE inside the unrolled stores they cost a
// scalar branch + s_waitcnt each (conv_halo64ws: 900 instructions, 122 branches between the last MFMA and the last store of a tile,
// ~1.9 us of a 7.6 us tile with the matrix pipe idle -- found with the RX_DBG=8 / 16 ablations, round 3).  RX_EPI_DISPATCH runs
// `body(HB, RA)` with the two flags as compile-time constants behind ONE uniform branch; the bias of a lane's 4 consecutive
// channels is one 16-byte load.
#define RX_EPI_DISPATCH(has_bias, rt_acc, body)                                                                     \
  do {                                                                                                              \
    if (!(has_bias) && !(rt_acc)) body(std::integral_constant<bool, false>{}, std::integral_constant<bool, false>{}); \
    else if ((has_bias) && !(rt_acc)) body(std::integral_constant<bool, true>{}, std::integral_constant<bool, false>{}); \
    else if (!(has_bias)) body(std::integral_constant<bool, false>{}, std::integral_constant<bool, true>{});          \
    else body(std::integral_constant<bool, true>{}, std::integral_constant<bool, true>{});                            \
  } while (0)

// ---- tap tables and MFMA wrappers shared by the implicit-GEMM and weight-gradient kernels ------
// kernel sizes 1..7 and strides 1..4 per axis (the reference passes any `kernel_sizes` / `strides` of a manual model_config
// straight to Conv(k, stride, pad = (k-1)//2), build_network_from_config.py:85-148): up to 7^3 = 343 taps per launch, so the
// weight index needs 9 bits and the tables live in the kernel arguments (RX_MAX_TAPS x 8 bytes, within the 4 KB kernarg limit)
#define RX_MAX_TAPS 344
#define RX_MAX_KERNEL 7
#define RX_MAX_STRIDE 4
struct RxTap {
  int8_t dz, dy, dx, pad_;
  uint16_t w, pad2_;
};

// one 32x32 accumulator update from two 16-byte operand fragments (lane = row/col (lane&31),
// k-half (lane>>5)): 16 k-values for 16-bit types, 8 for fp32 (4 exact-fp32 MFMAs).
template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  __device__ static inline void run(f32x16& c, const u32x4& a, const u32x4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<f16_t> {
  __device__ static inline void run(f32x16& c, const u32x4& a, const u32x4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  __device__ static inline void run(f32x16& c, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], c, 0, 0, 0);
  }
};

// ---- XCD-aware block order ------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (observed; a speed assumption only, never correctness), each with its
// own 4 MiB L2.  Remap the physical linear block id so that every XCD works through one CONTIGUOUS range of logical
// ids: blocks that share operand panels or halos then meet in the same L2.  Bijective for any grid size G.
__device__ inline int rx_xcd_remap(int L, int G) {
  const int x = L & 7, slot = L >> 3, chunk = G >> 3, rem = G & 7;
  return x * chunk + (x < rem ? x : rem) + slot;
}

// Walk order of the 256-voxel output tiles of the halo kernels (a speed choice only: any bijection is correct).
//   0  raster: x fastest.  A persistent workgroup that walks a contiguous range re-reads the y/z halo rows of every tile
//      from HBM (PMC: 2.46x the input tensor for the 4x4x16 tile, whose halo is 6x6x18) -- the reuse distance is a whole
//      x-row (y) or a whole plane (z) of tiles times the 32 workgroups that share the XCD's 4 MiB L2.
//   1  z fastest, then x, y, n: a persistent workgroup marches along z through a column of tiles, so its z-halo planes
//      are one tile old (L2 hit), and the workgroups of one XCD (contiguous ranges, rx_xcd_remap) march side by side
//      through neighbouring columns, so the x/y halos meet in that XCD's L2 at the same time.
//   2  4x4x4 bricks of tiles (needs tx_n, ty_n, tz_n % 4 == 0): for one-tile-per-workgroup grids, where the 64 workgroups an
//      XCD runs at a time should cover a compact brick (halo overhead 1.3x instead of 1.6x for a row or column of tiles).
__device__ __forceinline__ void rx_tile_coords(int t, int tx_n, int ty_n, int tz_n, int order, int& n, int& tz, int& ty, int& tx) {
  if (order == 1) {
    tz = t % tz_n;
    int c = t / tz_n;
    tx = c % tx_n, c /= tx_n;
    ty = c % ty_n, n = c / ty_n;
  } else if (order == 2) {
    int b = t >> 6;
    const int i = t & 63, bx_n = tx_n >> 2, by_n = ty_n >> 2, bz_n = tz_n >> 2;
    const int bx = b % bx_n;
    b /= bx_n;
    const int by = b % by_n;
    b /= by_n;
    const int bz = b % bz_n;
    n = b / bz_n;
    tx = bx * 4 + (i & 3), ty = by * 4 + ((i >> 2) & 3), tz = bz * 4 + (i >> 4);
  } else {
    tx = t % tx_n;
    int c = t / tx_n;
    ty = c % ty_n, c /= ty_n;
    tz = c % tz_n, n = c / tz_n;
  }
}

// ---- InstanceNorm statistics in the conv epilogue (persistent kernels, forward) --------------------------------------
// A consumer lane owns the same 16 (or 2 x 16) output channels for every voxel of every tile its workgroup walks, so the
// per-(n, c) sums of y and y^2 are RUNNING per-lane sums over the workgroup's whole life (2 VALU per stored value) and the
// cross-lane step -- a 5-step xor-shuffle over the 32 lanes of a half-wave, which own the same channels -- happens ONCE per
// workgroup, not per tile (round 1 rejected the per-tile version: ~320 shuffles per wave and tile).  Sums are taken of the
// values as STORED (rounded to the compute dtype): identical statistics to the separate pass over y.  Each consumer wave
// writes its own partial row; layout as colreduce_kernel's: part[((n*nchunks + chunk)*2 + a)*Co + c], chunk = wg*4 + wave.
template <int NA>
__device__ inline void ch_stat_flush(float (&s1)[NA][16], float (&s2)[NA][16], float* __restrict__ part, int n, int nchunks, int chunk, int Co,
                                     int n0, int lane) {
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float u = s1[a][r], v = s2[a][r];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        u += __shfl_xor(u, o, 64);
        v += __shfl_xor(v, o, 64);
      }
      s1[a][r] = u, s2[a][r] = v;
    }
  if ((lane & 31) == 0) {
    const int fh = lane >> 5;
    float* p0 = part + ((size_t)(n * nchunks + chunk) * 2) * Co + n0;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = a * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
        p0[co] = s1[a][r];
        p0[Co + co] = s2[a][r];
      }
  }
}
