// rx_pointwise.hip -- the convolutions WITHOUT a spatial footprint on one streaming kernel (16-bit types):
//   * nn.ConvTranspose3d with kernel == stride (decoder.py:110-113,146):   y[s*v + t][co] = sum_ci x[v][ci] W[t][co][ci] + b[co]
//   * 1x1x1 nn.Conv3d forward (skip projections, resblocks.py:89-104) and its data gradient (the same form on the swapped pack).
// The generic gather kernel (rx_igemm.hip) runs them as up to 8 single-tap GEMMs with a 64-byte K tile per barrier pair: at
// K = Ci = 64 that is two K steps per workgroup, and the 64 -> 32 transposed conv that writes the full-resolution concat half
// (268 MB) ran at 2.3 TB/s.  Here a wave owns 32 consecutive input voxels: their Ci channels are loaded ONCE, straight from
// global memory into the MFMA B fragments (16 bytes per lane and K step, no LDS), and stay in registers while the wave walks the
// taps; the weights of a 32-channel output block -- all taps, all Ci -- sit in LDS for the workgroup's whole life (rows padded
// by 16 bytes: 16 consecutive rows hit 16 distinct 4-bank groups).  Per tap: Ci/16 MFMAs 32x32x16, then the 32 x 32 tile leaves
// as 16-byte stores (rx_pair16).  Persistent over voxel tiles; HBM traffic = x once + y once.
#include "rx_common.h"

struct PwGeom {
  int N, Zi, Yi, Xi, Ci, ldi;
  long in_ss;
  int Yo, Xo, Co, ldo;
  long out_ss;
  int sz, sy, sx;      // output voxel = s * input voxel + tap (1 or 2 per axis); taps = sz * sy * sx
  int accumulate;
  int ntiles;          // tiles of 128 rows (n, z, y, x flattened)
};

#define RX_PW_MAXKS 16   // Ci <= 256: 16 B-fragment registers of 16 bytes per lane

template <typename T>
__global__ __launch_bounds__(256) void pointwise_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, const PwGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [taps][32 co][Ci * 2 + 16 bytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntaps = g.sz * g.sy * g.sx;
  const int pitch = g.Ci * 2 + 16;
  const int n0 = blockIdx.y * 32;
  const int cpr = g.Ci / 8;                           // 16-byte chunks per weight row
  for (int i = tid; i < ntaps * 32 * cpr; i += 256) {
    const int c = i % cpr, row = i / cpr;             // row = t * 32 + co
    const int t = row >> 5, co = row & 31;
    *reinterpret_cast<u32x4*>(smem + row * pitch + c * 16) =
        *reinterpret_cast<const u32x4*>(w + ((long)t * g.Co + n0 + co) * g.Ci + c * 8);
  }
  __syncthreads();
  const int KS = g.Ci >> 4;
  const long V = (long)g.Zi * g.Yi * g.Xi, NV = (long)g.N * V;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[n0 + 8 * (r >> 2) + 4 * fh + (r & 3)] : 0.f;
  const unsigned char* wrow = smem + fr * pitch + fh * 16;
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const long q = (long)tile * 128 + wave * 32 + fr;
    const bool ok = q < NV;
    const int n = ok ? (int)(q / V) : 0;
    const long v = ok ? q - (long)n * V : 0;
    const int x = (int)(v % g.Xi), y = (int)((v / g.Xi) % g.Yi), z = (int)(v / ((long)g.Xi * g.Yi));
    u32x4 bq[RX_PW_MAXKS];
    const T* ip = in + (long)n * g.in_ss + v * g.ldi + fh * 8;
#pragma unroll
    for (int ks = 0; ks < RX_PW_MAXKS; ++ks)
      if (ks < KS) bq[ks] = ok ? *reinterpret_cast<const u32x4*>(ip + ks * 16) : u32x4{0u, 0u, 0u, 0u};
    T* on = out + (long)n * g.out_ss + n0;
    for (int t = 0; t < ntaps; ++t) {
      const int c = t % g.sx, b = (t / g.sx) % g.sy, a = t / (g.sx * g.sy);
      const long ov = ((long)(z * g.sz + a) * g.Yo + (y * g.sy + b)) * g.Xo + (x * g.sx + c);
      T* op = on + ov * g.ldo;
      u32x2 oldv[4];
      if (g.accumulate && ok) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) oldv[g4] = *reinterpret_cast<const u32x2*>(op + 8 * g4 + 4 * fh);
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const unsigned char* wt = wrow + t * 32 * pitch;
#pragma unroll
      for (int ks = 0; ks < RX_PW_MAXKS; ++ks)
        if (ks < KS) Mma<T>::run(acc, *reinterpret_cast<const u32x4*>(wt + ks * 32), bq[ks]);
      u32x2 piece[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        T vals[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float f = acc[4 * g4 + i] + bv[4 * g4 + i];
          if (g.accumulate) f += Elem<T>::to_f(reinterpret_cast<const T*>(&oldv[g4])[i]);
          vals[i] = Elem<T>::from_f(f);
        }
        piece[g4] = *reinterpret_cast<u32x2*>(vals);
      }
      // lanes l and l + 32 hold the same voxel (both pass `ok` together): pair their 8-byte pieces into 16-byte stores
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u32x4 o16 = rx_pair16(piece[2 * pr], piece[2 * pr + 1]);
        if (ok) *reinterpret_cast<u32x4*>(op + 16 * pr + 8 * fh) = o16;
      }
    }
  }
}

// 1 = handled.  in: the tensor on the small grid (Zi, Yi, Xi); out voxel = s * in voxel + tap; w = [taps][Co][Ci] packed weights.
int rx_pointwise_try(rx_dtype dt, const rx_act* in, const void* w, const float* bias, const rx_act* out, const int32_t stride[3],
                     int accumulate, hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("RX_POINTWISE");
    on = e ? atoi(e) : 1;
  }
  if (!on || dt == RX_F32 || in->cs || out->cs) return 0;
  const int taps = stride[0] * stride[1] * stride[2];
  const int Ci = in->c, Co = out->c;
  if (Ci % 16 || Ci > 16 * RX_PW_MAXKS || Co % 32 || in->ld % 8 || out->ld % 8 || ((uintptr_t)in->ptr & 15) || ((uintptr_t)out->ptr & 15) ||
      ((uintptr_t)w & 15))
    return 0;
  const size_t lds = (size_t)taps * 32 * (Ci * 2 + 16);
  if (lds > 150 * 1024) return 0;
  if (taps > 1 && lds > 80 * 1024) return 0;          // one workgroup per CU: the 256 -> 128 transposed conv measured 20.7 vs 19.1 us on the gather kernel
  const long NV = (long)in->n * rx_act_voxels(in);
  if (NV < 4096) return 0;                            // the low-resolution layers stay on the split-K kernels
  PwGeom g;
  g.N = in->n, g.Zi = in->z, g.Yi = in->y, g.Xi = in->x, g.Ci = Ci, g.ldi = in->ld;
  g.in_ss = rx_act_voxels(in) * (long)in->ld;
  g.Yo = out->y, g.Xo = out->x, g.Co = Co, g.ldo = out->ld;
  g.out_ss = rx_act_voxels(out) * (long)out->ld;
  g.sz = stride[0], g.sy = stride[1], g.sx = stride[2];
  g.accumulate = accumulate;
  g.ntiles = (int)((NV + 127) / 128);
  // persistent: as many workgroups as stay resident (LDS-limited), each walking tiles with stride gridDim.x
  int per_cu = (int)((160 * 1024) / (lds + 1024));
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int gx = 256 * per_cu / (Co / 32);
  if (gx < 1) gx = 1;
  if (gx > g.ntiles) gx = g.ntiles;
  dim3 grid(gx, Co / 32);
  rx_note_kernel("pointwise_kernel");
  if (dt == RX_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pointwise_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((pointwise_kernel<bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)in->ptr, (const bf16_t*)w, bias, (bf16_t*)out->ptr, g);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pointwise_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((pointwise_kernel<f16_t>), grid, dim3(256), lds, st, (const f16_t*)in->ptr, (const f16_t*)w, bias, (f16_t*)out->ptr, g);
  }
  return 1;
}

