// rx_conv_halo.hip -- forward / backward-data of the stride-1 3x3x3 convolutions with an LDS halo tile.
//
//   Out[v][co] (+)= sum_t sum_ci W[t][co][ci] * In[v + d_t][ci] (+ bias),   d_t = t-1 (fwd) or 1-t (bwd-data)
//
// The generic gather kernel (rx_igemm.hip) re-stages a [voxels][64 B] activation tile for every (tap, channel
// chunk): 27 global->LDS round trips per chunk, each only 8 MFMAs deep -> latency bound (~17 % of MFMA peak).
// Here a workgroup owns an output tile of TZ x TY x TX = 256 voxels x BN channels and, per 64-byte input-channel
// chunk, stages the (TZ+2)(TY+2)(TX+2) halo rows ONCE; all 27 taps read it with a row offset.  Weights are
// streamed in three dz-planes of 9 taps ([9][BN][64 B] in LDS).  Phases (chunk, dz-plane) are software
// pipelined: the global loads of phase p+1 are in flight in registers while phase p runs 36..72 MFMAs per wave,
// and two workgroups share a CU (<= 80 KB LDS each).
// MFMA orientation and epilogue are those of rx_igemm.hip (weights = A operand, voxels = accumulator lanes).
#include <stdlib.h>

#include <type_traits>

#include "rx_common.h"

struct ConvHaloGeom {
  int N, Z, Y, X;
  int Ci, Co, ldi, ldo;
  long in_ss, out_ss;
  long in_cs, out_cs;         // element offset between consecutive 32-channel groups of in / out (planar concat: rx_act.cs)
  int TZ, TY, TX, lTX, lTY;
  int HY, HX, HV, VT;
  int tz_n, ty_n, tx_n, NT;
  int order;                  // tile walk order (rx_tile_coords)
  int wgs_s;                  // persistent kernels: workgroups per SAMPLE (tile ranges never straddle samples); 0 = off
  float* stat_part;           // fused InstanceNorm statistics: per-wave partial sums, or nullptr (see ch_stat_flush)
  // conv_halo32p<.., BS>: backward-data launches that COMPLETE the gradient of an InstanceNorm layer's output also accumulate
  // that layer's two backward sums  s1 = sum g',  s2 = sum g'*(y - mean)  (g' = g * lrelu'(xhat): layers without a residual,
  // whose mask is the sign of y - mean) the same way -- y prefetched under the MFMA loop.  bs_y / bs_stats describe the layer.
  const void* bs_y;
  const float* bs_stats;
  int bs_ldy;
  long bs_yss;
  float bs_slope;
  int accumulate, flip, dbg;  // dbg: ablation mask (RX_DBG env): 1 no halo loads, 2 no weight loads, 4 no MFMA, 8 no stores
};

#define RX_CH_MAX_HV 656   // (4,4,16): 648 rows, (4,8,8)/(8,8,4): 600; 2 workgroups of <= 79 KB per CU
#define RX_CH_XPIECES ((RX_CH_MAX_HV * 4 + 255) / 256)

__device__ inline int hswz(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }

// ds_read_b128 is serviced in four 16-lane groups that are NOT contiguous lane ranges (MI355X_MICROARCH.md, LDS):
// {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32.  The accumulator column a lane owns (lane & 31) can be any
// voxel of the wave's 32-voxel block as long as the B-fragment read and the epilogue agree, so voxels are dealt to
// lanes such that each hardware group covers 16 CONSECUTIVE voxels of one x-row: with 80-byte rows (or 64-byte rows
// XOR-swizzled on (row>>2)&3) 16 consecutive rows are bank-conflict free.
__device__ inline int lane_voxel(int l /* lane & 31 */) {
  return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
}

template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                           T* __restrict__ out, const ConvHaloGeom g, float* __restrict__ slab, int cps) {
  constexpr int P = Elem<T>::PER16;
  constexpr int KB = 4 * P;             // input channels per 64-byte chunk
  constexpr int NB = BN / 32;
  constexpr int MV = 2;                 // voxel blocks per wave (VT = 256)
  constexpr int WPIECES = (9 * BN * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* sX = reinterpret_cast<u32x4*>(smem);                   // [HV][4 chunks]
  u32x4* sW = sX + RX_CH_MAX_HV * 4;                            // [9][BN][4 chunks]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // logical id: every XCD works through a contiguous range of tiles, the channel blocks of a tile back to back
  const int lid = g.order ? rx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) : blockIdx.x * gridDim.y + blockIdx.y;
  const int tile = lid / gridDim.y, n0 = (lid - tile * gridDim.y) * BN;
  int tx, ty, tz, n;
  rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
  const int z0 = tz * g.TZ, y0 = ty * g.TY, x0 = tx * g.TX;
  const T* in_n = in + (long)n * g.in_ss;

  // ---- staging geometry of this thread: halo rows (tid>>2)+64p, chunk tid&3; global offset or -1 (zero fill)
  const int chunk = tid & 3;
  int xoff[RX_CH_XPIECES];
#pragma unroll
  for (int p = 0; p < RX_CH_XPIECES; ++p) {
    int row = (tid >> 2) + 64 * p;
    xoff[p] = -2;  // not a halo row
    if (row < g.HV) {
      int hx = row % g.HX, t = row / g.HX;
      int hy = t % g.HY, hz = t / g.HY;
      int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
      bool ok = (unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X;
      xoff[p] = ok ? (int)(((long)(z * g.Y + y) * g.X + x) * g.ldi) + chunk * P : -1;
    }
  }
  // split-K (slab != nullptr; the 8^3 layers, whose 32 (tile, channel block) pairs cannot fill 256 CUs): blockIdx.z owns the
  // 64-byte input-channel chunks [cbeg, cend) and leaves an fp32 partial tile in slab[split][n*V + v][co] (ch_splitk_reduce)
  const int nchunks = g.Ci / KB;
  const int cbeg = slab ? blockIdx.z * cps : 0, cend = slab ? min(nchunks, cbeg + cps) : nchunks;
  const int nphase = (cend - cbeg) * 3;

  u32x4 xr[RX_CH_XPIECES], wr[WPIECES];
  auto prefetch = [&](int ph) {
    const int cc = cbeg + ph / 3, dzg = ph % 3;
    if (dzg == 0) {
#pragma unroll
      for (int p = 0; p < RX_CH_XPIECES; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (xoff[p] >= 0) v = *reinterpret_cast<const u32x4*>(in_n + xoff[p] + cc * KB);
        xr[p] = v;
      }
    }
    // weights of taps 9*dzg .. 9*dzg+8 (flip: the plane of taps whose dz offset is the same)
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      int i = tid + 256 * p;          // piece index: ((tl*BN + row)*4 + chunk)
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (i < 9 * BN * 4) {
        int c4 = i & 3, r = (i >> 2) % BN, tl = (i >> 2) / BN;
        int t = 9 * dzg + tl;
        v = *reinterpret_cast<const u32x4*>(w + ((long)t * g.Co + n0 + r) * g.Ci + cc * KB + c4 * P);
      }
      wr[p] = v;
    }
  };
  auto commit = [&](int ph) {
    const int dzg = ph % 3;
    if (dzg == 0) {
#pragma unroll
      for (int p = 0; p < RX_CH_XPIECES; ++p) {
        int row = (tid >> 2) + 64 * p;
        if (xoff[p] != -2) sX[hswz(row, chunk)] = xr[p];
      }
    }
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      int i = tid + 256 * p;
      if (i < 9 * BN * 4) {
        int c4 = i & 3, rr = i >> 2;  // rr = tl*BN + r
        sW[hswz(rr, c4)] = wr[p];
      }
    }
  };

  // ---- fragment geometry: this lane's voxel in each of its MV blocks -> halo row
  const int fr = lane & 31, fh = lane >> 5;
  // TX == 16: deal voxels to lanes so that each ds_read_b128 lane group reads 16 consecutive (swizzle-conflict-free) rows
  const int fv = g.TX == 16 ? lane_voxel(fr) : fr;
  int hrow[MV];
#pragma unroll
  for (int b = 0; b < MV; ++b) {
    int v = (wave * MV + b) * 32 + fv;
    int vx = v & (g.TX - 1), vy = (v >> g.lTX) & (g.TY - 1), vz = v >> (g.lTX + g.lTY);
    hrow[b] = ((vz + 1) * g.HY + (vy + 1)) * g.HX + vx + 1;
  }
  const int sgn = g.flip ? -1 : 1;

  f32x16 acc[NB][MV];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < MV; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  prefetch(0);
  for (int ph = 0; ph < nphase; ++ph) {
    __syncthreads();  // previous phase fully consumed
    commit(ph);
    __syncthreads();
    if (ph + 1 < nphase) prefetch(ph + 1);
    const int dzg = ph % 3;
    const int dz = sgn * (dzg - 1);
#pragma unroll
    for (int tl = 0; tl < 9; ++tl) {
      const int dy = sgn * (tl / 3 - 1), dx = sgn * (tl % 3 - 1);
      const int toff = (dz * g.HY + dy) * g.HX + dx;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 af[NB], bf[MV];
#pragma unroll
        for (int a = 0; a < NB; ++a) af[a] = sW[hswz(tl * BN + a * 32 + fr, ks * 2 + fh)];
#pragma unroll
        for (int b = 0; b < MV; ++b) bf[b] = sX[hswz(hrow[b] + toff, ks * 2 + fh)];
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
          for (int b = 0; b < MV; ++b) Mma<T>::run(acc[a][b], af[a], bf[b]);
      }
    }
  }

  // ---- epilogue (as rx_igemm.hip): lane = voxel, 4 runs of 4 consecutive channels per 32-block
#pragma unroll
  for (int b = 0; b < MV; ++b) {
    int v = (wave * MV + b) * 32 + fv;
    int vx = v & (g.TX - 1), vy = (v >> g.lTX) & (g.TY - 1), vz = v >> (g.lTX + g.lTY);
    int z = z0 + vz, y = y0 + vy, x = x0 + vx;
    if (z >= g.Z || y >= g.Y || x >= g.X) continue;
    if (slab) {
      float* sp = slab + (((long)blockIdx.z * g.N + n) * ((long)g.Z * g.Y * g.X) + ((long)(z * g.Y + y) * g.X + x)) * g.Co + n0;
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<f32x4*>(sp + a * 32 + 8 * g4 + 4 * fh) =
              f32x4{acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
      continue;
    }
    T* op = out + (long)n * g.out_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldo + n0;
    auto epi = [&](auto HB, auto RA) {
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int co = a * 32 + 8 * g4 + 4 * fh;
          f32x4 bv = {0.f, 0.f, 0.f, 0.f};
          if (decltype(HB)::value) bv = *reinterpret_cast<const f32x4*>(bias + n0 + co);
          T old4[4];
          if (decltype(RA)::value) {
            if (sizeof(T) == 2) *reinterpret_cast<u32x2*>(old4) = *reinterpret_cast<const u32x2*>(op + co);
            else *reinterpret_cast<u32x4*>(old4) = *reinterpret_cast<const u32x4*>(op + co);
          }
          T vals[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float f = acc[a][b][4 * g4 + i] + bv[i];
            if (decltype(RA)::value) f += Elem<T>::to_f(old4[i]);
            vals[i] = Elem<T>::from_f(f);
          }
          if (sizeof(T) == 2)
            *reinterpret_cast<u32x2*>(op + co) = *reinterpret_cast<u32x2*>(vals);
          else
            *reinterpret_cast<u32x4*>(op + co) = *reinterpret_cast<u32x4*>(vals);
        }
    };
    RX_EPI_DISPATCH(bias != nullptr, g.accumulate != 0, epi);
  }
}

// out[row][c] (+)= sum_s slab[s][row][c] (+ bias): fixed order, one thread per 4 channels of a voxel row (row = n*V + v)
template <typename T>
__global__ __launch_bounds__(256) void ch_splitk_reduce(const float* __restrict__ slab, int S, long rows, int Co, const float* __restrict__ bias,
                                                        T* __restrict__ out, int ldo, int accumulate) {
  const int CV = Co / 4;
  const long total = rows * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const long row = i / CV;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < S; ++k) a += *reinterpret_cast<const f32x4*>(slab + ((long)k * rows + row) * Co + cv * 4);
    T* op = out + row * ldo + cv * 4;
    T vals[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = a[j];
      if (bias) f += bias[cv * 4 + j];
      if (accumulate) f += Elem<T>::to_f(op[j]);
      vals[j] = Elem<T>::from_f(f);
    }
    if (sizeof(T) == 2)
      *reinterpret_cast<u32x2*>(op) = *reinterpret_cast<u32x2*>(vals);
    else
      *reinterpret_cast<u32x4*>(op) = *reinterpret_cast<u32x4*>(vals);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Specialisation for the full-resolution layers (Co block of 32, tile 4x4x16): everything about the tile is a
// compile-time constant, halo rows are padded to 80 bytes (16 consecutive rows hit 16 distinct 16-byte bank
// slots: (20*r mod 64) is a bijection on r mod 16) so the LDS address of (voxel, tap, k-step) is
// lane_base + CONSTANT and folds into the ds_read_b128 offset field -- no address arithmetic in the MFMA loop
// (the generic kernel above spends ~14 VALU instructions per MFMA on XOR-swizzled addresses).
// ---------------------------------------------------------------------------------------------------------
template <int NA>
__device__ inline void ch_stat_flush(float (&s1)[NA][16], float (&s2)[NA][16], float* __restrict__ part, int n, int nchunks, int chunk, int Co,
                                     int n0, int lane);      // (defined with the persistent kernels below)

template <typename T, bool FLIP, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv_halo32_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                             T* __restrict__ out, const ConvHaloGeom g) {
  constexpr int P = Elem<T>::PER16;
  constexpr int KB = 4 * P;
  constexpr int TZ = 4, TY = 4, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;  // 648 rows
  constexpr int ROWB = 80;                     // bytes per halo row (64 payload + 16 pad)
  constexpr int XPIECES = (HV * 4 + 255) / 256;  // 11
  constexpr int WPIECES = (9 * 32 * 4 + 255) / 256;  // 5
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;                    // [HV][80 B]
  unsigned char* sW = smem + HV * ROWB;        // [9][32][64 B], chunk XOR (row>>2)&3

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = g.order ? rx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) : blockIdx.x * gridDim.y + blockIdx.y;
  const int tile = lid / gridDim.y, n0 = (lid - tile * gridDim.y) * 32;
  int tx, ty, tz, n;
  rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
  const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
  const T* in_n = in + (long)n * g.in_ss;
  const int chunk = tid & 3;

  int xoff[XPIECES];  // element offset of this thread's halo pieces, -1 = zero fill, -2 = no such row
#pragma unroll
  for (int p = 0; p < XPIECES; ++p) {
    const int row = (tid >> 2) + 64 * p;
    xoff[p] = -2;
    if (row < HV) {
      const int hx = row % HX, t = row / HX, hy = t % HY, hz = t / HY;
      const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
      const bool ok = (unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X;
      xoff[p] = ok ? (int)(((long)(z * g.Y + y) * g.X + x) * g.ldi) + chunk * P : -1;
    }
  }
  const int nchunks = g.Ci / KB, nphase = nchunks * 3;
  u32x4 xr[XPIECES], wr[WPIECES];
  auto prefetch = [&](int ph) {
    const int cc = ph / 3, dzg = ph - cc * 3;
    if (dzg == 0) {
#pragma unroll
      for (int p = 0; p < XPIECES; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (xoff[p] >= 0 && !RX_ABLATE(g, 1)) v = *reinterpret_cast<const u32x4*>(in_n + xoff[p] + cc * g.in_cs);
        xr[p] = v;
      }
    }
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      const int i = tid + 256 * p;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (i < 9 * 32 * 4 && !RX_ABLATE(g, 2)) {
        const int c4 = i & 3, r = (i >> 2) & 31, tl = i >> 7;
        v = *reinterpret_cast<const u32x4*>(w + ((long)(9 * dzg + tl) * g.Co + n0 + r) * g.Ci + cc * KB + c4 * P);
      }
      wr[p] = v;
    }
  };
  auto commit = [&](int ph) {
    if (ph % 3 == 0) {
#pragma unroll
      for (int p = 0; p < XPIECES; ++p) {
        const int row = (tid >> 2) + 64 * p;
        if (xoff[p] != -2) *reinterpret_cast<u32x4*>(sX + row * ROWB + chunk * 16) = xr[p];
      }
    }
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      const int i = tid + 256 * p;
      if (i < 9 * 32 * 4) {
        const int c4 = i & 3, rr = i >> 2;
        *reinterpret_cast<u32x4*>(sW + rr * 64 + ((c4 ^ ((rr >> 2) & 3)) << 4)) = wr[p];
      }
    }
  };

  // lane bases (bytes).  voxel v = (wave*2+b)*32 + fr, x fastest: vx = v&15, vy = (v>>4)&3, vz = v>>6
  const int fr = lane & 31, fh = lane >> 5;
  const int fv = lane_voxel(fr);                         // voxel (within a 32-block) whose accumulator column this lane owns
  const unsigned char* xb[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int v = (wave * 2 + b) * 32 + fv;
    const int vx = v & 15, vy = (v >> 4) & 3, vz = v >> 6;
    xb[b] = sX + (((vz + 1) * HY + (vy + 1)) * HX + vx + 1) * ROWB + fh * 16;
  }
  const int wsw = (fr >> 2) & 3;                         // weight-row swizzle term of this lane
  const unsigned char* wb0 = sW + fr * 64 + (((0 + fh) ^ wsw) << 4);  // k-step 0
  const unsigned char* wb1 = sW + fr * 64 + (((2 + fh) ^ wsw) << 4);  // k-step 1
  constexpr int sgn = FLIP ? -1 : 1;                   // compile-time: the tap offsets stay ds_read immediates

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  prefetch(0);
  for (int ph = 0; ph < nphase; ++ph) {
    __syncthreads();
    commit(ph);
    __syncthreads();
    if (ph + 1 < nphase) prefetch(ph + 1);
    const int dzoff = sgn * ((ph % 3) - 1) * (HY * HX * ROWB);  // wave-uniform byte offset of this dz plane
    const unsigned char* x0p = xb[0] + dzoff;
    const unsigned char* x1p = xb[1] + dzoff;
    if (!RX_ABLATE(g, 4)) {
      // 18 steps (9 taps x 2 k-steps), software pipelined by hand: the three fragment reads of step i+1 are issued
      // before the two MFMAs of step i; sched_group_barrier pins the 2-MFMA / 3-DS_READ interleave.
      constexpr int ssg = sgn;
      u32x4 fa[2], f0[2], f1[2];
      auto ld = [&](int i, int slot) {
        const int tl = i >> 1, ks = i & 1;
        const int toff = ssg * ((tl / 3 - 1) * HX + (tl % 3 - 1)) * ROWB;
        fa[slot] = *reinterpret_cast<const u32x4*>((ks ? wb1 : wb0) + tl * 32 * 64);
        f0[slot] = *reinterpret_cast<const u32x4*>(x0p + toff + ks * 32);
        f1[slot] = *reinterpret_cast<const u32x4*>(x1p + toff + ks * 32);
      };
      ld(0, 0);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        if (i + 1 < 18) ld(i + 1, (i + 1) & 1);
        Mma<T>::run(acc[0], fa[i & 1], f0[i & 1]);
        Mma<T>::run(acc[1], fa[i & 1], f1[i & 1]);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // 3 DS reads (next step)
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMAs (this step)
      }
    }
  }

  float st1[1][16], st2[1][16];
  if (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st1[0][r] = 0.f, st2[0][r] = 0.f;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int v = (wave * 2 + b) * 32 + fv;
    const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
    if (z >= g.Z || y >= g.Y || x >= g.X || RX_ABLATE(g, 8)) continue;
    T* op = out + (long)n * g.out_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldo + n0;
    auto epi = [&](auto HB, auto RA) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int co = 8 * g4 + 4 * fh;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (decltype(HB)::value) bv = *reinterpret_cast<const f32x4*>(bias + n0 + co);
        T old4[4];
        if (decltype(RA)::value) {
          if (sizeof(T) == 2) *reinterpret_cast<u32x2*>(old4) = *reinterpret_cast<const u32x2*>(op + co);
          else *reinterpret_cast<u32x4*>(old4) = *reinterpret_cast<const u32x4*>(op + co);
        }
        T vals[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float f = acc[b][4 * g4 + i] + bv[i];
          if (decltype(RA)::value) f += Elem<T>::to_f(old4[i]);
          vals[i] = Elem<T>::from_f(f);
          if (STATS) {
            const float r = Elem<T>::to_f(vals[i]);
            st1[0][4 * g4 + i] += r;
            st2[0][4 * g4 + i] += r * r;
          }
        }
        if (sizeof(T) == 2)
          *reinterpret_cast<u32x2*>(op + co) = *reinterpret_cast<u32x2*>(vals);
        else
          *reinterpret_cast<u32x4*>(op + co) = *reinterpret_cast<u32x4*>(vals);
      }
    };
    RX_EPI_DISPATCH(bias != nullptr, g.accumulate != 0, epi);
  }
  // STATS (deep layers only: one tile per workgroup, but hundreds of MFMAs per wave behind each wavefront reduction): the
  // InstanceNorm sums of this wave's 64 voxels, partial row (tile of the sample, wave)
  if (STATS) {
    const int NTs = g.NT / g.N;
    ch_stat_flush<1>(st1, st2, g.stat_part, n, NTs * 4, (tile - n * NTs) * 4 + wave, g.Co, n0, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent, weight-stationary, wave-specialised variant for the 32 -> 32 channel layers at full resolution (16-bit
// types: one 64-byte channel chunk).  Ablation of conv_halo32 on 32->32 @128^3 (366 us): without MFMAs 224 us -- per
// 256-voxel tile a workgroup re-stages ALL 27x32x32 weights (55 KB) next to a 41 KB halo for only 108 MFMAs per wave.
// Here a workgroup keeps the weights in LDS for its whole life and walks a contiguous range of tiles:
//   LDS = 27x32 weight rows (55,296 B) + TWO halo buffers of 648 x 80 B (103,680 B) = 158,976 B -> one workgroup per CU,
//   512 threads = 4 consumer waves (MFMA loop of conv_halo32 + epilogue stores) and 4 producer waves (halo of the NEXT
//   tile: global -> registers -> LDS), one barrier per tile.
// ---------------------------------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. makes every wave wait for the
// write acknowledgements of its epilogue stores (ablation: 1.3 us per tile) although nobody in the kernel reads them
__device__ inline void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// (ch_stat_flush: rx_common.h)

#define CH32P_HALO_BYTES (648 * 80)
#define CH32P_W_BYTES (27 * 32 * 64)

// STATS: forward instantiation that accumulates the InstanceNorm statistics of its output; ACC: dx += (old values prefetched
// under the MFMA loop).  Compile-time so that the plain variant keeps its 103 registers: with run-time flags the statistics
// code alone cost every launch 10-17 % (306 -> 358 us for the 32 -> 32 data gradient @128^3).
template <typename T, bool STATS = false, bool ACC = false, bool FLIP = false, bool BS = false>
__global__ __launch_bounds__(512, 1) void conv_halo32p_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                              T* __restrict__ out, const ConvHaloGeom g, int tiles_per_wg) {
  constexpr int P = Elem<T>::PER16;
  constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2, HV = 648;
  constexpr int ROWB = 80;
  constexpr int XPIECES = (HV * 4 + 255) / 256;  // 11 (256 producer threads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;                            // [27][32][64 B], chunk XOR (row>>2)&3
  unsigned char* sX0 = smem + CH32P_W_BYTES;           // two halo buffers [HV][80 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = g.order ? rx_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  int t_begin = vb * tiles_per_wg, t_end = min(g.NT, t_begin + tiles_per_wg);
  const int NTs = g.NT / g.N;                          // tiles per sample
  const int sn = g.wgs_s ? vb / g.wgs_s : 0, sl = g.wgs_s ? vb - sn * g.wgs_s : 0;
  if (g.wgs_s) t_begin = sn * NTs + sl * tiles_per_wg, t_end = min((sn + 1) * NTs, t_begin + tiles_per_wg);

  // ---- weights: once per workgroup (all 512 threads)
  for (int i = tid; i < 27 * 32 * 4; i += 512) {
    const int c4 = i & 3, rr = i >> 2;                 // rr = tap*32 + co
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)rr * g.Ci + c4 * P);
    *reinterpret_cast<u32x4*>(sW + rr * 64 + ((c4 ^ ((rr >> 2) & 3)) << 4)) = v;
  }

  auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
    int tx, ty, tz;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
    z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
  };

  if (wave >= 4) {
    // ================================= producers =================================
    const int ptid = tid - 256, chunk = ptid & 3;
    int xh[XPIECES];
#pragma unroll
    for (int p = 0; p < XPIECES; ++p) {
      const int row = (ptid >> 2) + 64 * p;
      const int hx = row % HX, t = row / HX;
      xh[p] = row < HV ? ((t / HY) << 16) | ((t % HY) << 8) | hx : -1;
    }
    u32x4 xr[XPIECES];
    auto load_halo = [&](int tile) {
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      const T* in_n = in + (long)n * g.in_ss + chunk * P;
#pragma unroll
      for (int p = 0; p < XPIECES; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (xh[p] >= 0) {
          const int z = z0 + (xh[p] >> 16) - 1, y = y0 + ((xh[p] >> 8) & 255) - 1, x = x0 + (xh[p] & 255) - 1;
          if ((unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X && !RX_ABLATE(g, 1))
            v = *reinterpret_cast<const u32x4*>(in_n + ((long)(z * g.Y + y) * g.X + x) * g.ldi);
        }
        xr[p] = v;
      }
    };
    auto commit = [&](int buf) {
      unsigned char* sX = sX0 + buf * CH32P_HALO_BYTES;
#pragma unroll
      for (int p = 0; p < XPIECES; ++p) {
        const int row = (ptid >> 2) + 64 * p;
        if (xh[p] >= 0) *reinterpret_cast<u32x4*>(sX + row * ROWB + chunk * 16) = xr[p];
      }
    };
    if (t_begin < t_end) {
      load_halo(t_begin);
      commit(0);
      if (t_begin + 1 < t_end) load_halo(t_begin + 1);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int buf = (tile - t_begin) & 1;
      if (tile + 1 < t_end) {
        commit(buf ^ 1);
        if (tile + 2 < t_end) load_halo(tile + 2);
      }
      lds_only_barrier();
    }
  } else {
    // ================================= consumers =================================
    const int fr = lane & 31, fh = lane >> 5;
    const int fv = lane_voxel(fr);
    int xoffb[2];                                       // byte offset of this lane's voxel centre inside a halo buffer
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int v = (wave * 2 + b) * 32 + fv;
      const int vx = v & 15, vy = (v >> 4) & 3, vz = v >> 6;
      xoffb[b] = (((vz + 1) * HY + (vy + 1)) * HX + vx + 1) * ROWB + fh * 16;
    }
    const int wsw = (fr >> 2) & 3;
    const unsigned char* wb0 = sW + fr * 64 + (((0 + fh) ^ wsw) << 4);
    const unsigned char* wb1 = sW + fr * 64 + (((2 + fh) ^ wsw) << 4);
    constexpr int sgn = FLIP ? -1 : 1;                 // compile-time: the tap offsets stay ds_read immediates
    float s1[1][16], s2[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[0][r] = 0.f, s2[0][r] = 0.f;
    float bvp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bvp[r] = bias ? bias[8 * (r >> 2) + 4 * fh + (r & 3)] : 0.f;
    float bmean[16];
    if (BS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bmean[r] = g.bs_stats[2 * ((size_t)sn * g.Co + 8 * (r >> 2) + 4 * fh + (r & 3))];
    }
    __syncthreads();                                    // weights + tile 0
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int buf = (tile - t_begin) & 1;
      const unsigned char* sX = sX0 + buf * CH32P_HALO_BYTES;
      // accumulate (dx += ...): the old values are fetched HERE, under the MFMA loop -- read in the epilogue they cost a full
      // memory round trip per tile (32->32 dgrad @128^3: 306 us without, 540 us with accumulation)
      u32x2 oldv[2][4], yq[2][4];
      if (ACC || BS) {
        int n_, z0_, y0_, x0_;
        tile_origin(tile, n_, z0_, y0_, x0_);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int v = (wave * 2 + b) * 32 + fv;
          const int z = z0_ + (v >> 6), y = y0_ + ((v >> 4) & 3), x = x0_ + (v & 15);
          const bool ok = z < g.Z && y < g.Y && x < g.X;
          const long vox = (long)(z * g.Y + y) * g.X + x;
          if (ACC) {
            const T* op = out + (long)n_ * g.out_ss + vox * g.ldo + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) oldv[b][g4] = ok ? *reinterpret_cast<const u32x2*>(op + 8 * g4) : u32x2{0u, 0u};
          }
          if (BS) {
            const T* yp = (const T*)g.bs_y + (long)n_ * g.bs_yss + vox * g.bs_ldy + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) yq[b][g4] = ok ? *reinterpret_cast<const u32x2*>(yp + 8 * g4) : u32x2{0u, 0u};
          }
        }
      }
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      if (!RX_ABLATE(g, 4))
#pragma unroll
      for (int dzg = 0; dzg < 3; ++dzg) {
        const int dzoff = sgn * (dzg - 1) * (HY * HX * ROWB);
        const unsigned char* x0p = sX + xoffb[0] + dzoff;
        const unsigned char* x1p = sX + xoffb[1] + dzoff;
        const unsigned char* w0 = wb0 + dzg * 9 * 32 * 64;
        const unsigned char* w1 = wb1 + dzg * 9 * 32 * 64;
        // one consumer wave per SIMD: nobody else hides the LDS latency, so fragments are read DEPTH steps (2*DEPTH MFMAs)
        // ahead into a rolling ring of register slots, pinned with sched_barrier
        constexpr int DEPTH = 3;
        u32x4 fa[DEPTH + 1], f0[DEPTH + 1], f1[DEPTH + 1];
        auto ld = [&](int i, int slot) {
          const int tl = i >> 1, ks = i & 1;
          const int toff = sgn * ((tl / 3 - 1) * HX + (tl % 3 - 1)) * ROWB;
          fa[slot] = *reinterpret_cast<const u32x4*>((ks ? w1 : w0) + tl * 32 * 64);
          f0[slot] = *reinterpret_cast<const u32x4*>(x0p + toff + ks * 32);
          f1[slot] = *reinterpret_cast<const u32x4*>(x1p + toff + ks * 32);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ld(i, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          if (i + DEPTH < 18) ld(i + DEPTH, (i + DEPTH) % (DEPTH + 1));
          __builtin_amdgcn_sched_barrier(0);
          Mma<T>::run(acc[0], fa[i % (DEPTH + 1)], f0[i % (DEPTH + 1)]);
          Mma<T>::run(acc[1], fa[i % (DEPTH + 1)], f1[i % (DEPTH + 1)]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- epilogue of this tile
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int v = (wave * 2 + b) * 32 + fv;
        const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
        if (z >= g.Z || y >= g.Y || x >= g.X || RX_ABLATE(g, 8)) continue;
        T* op = out + (long)n * g.out_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldo;
        u32x2 piece[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int co = 8 * g4 + 4 * fh;
          T vals[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float f = acc[b][4 * g4 + i] + bvp[4 * g4 + i];      // (bias of this lane's 16 channels, loaded once per workgroup)
            if (ACC) f += Elem<T>::to_f(reinterpret_cast<const T*>(&oldv[b][g4])[i]);
            vals[i] = Elem<T>::from_f(f);
            if (STATS) {
              const float r = Elem<T>::to_f(vals[i]);
              s1[0][4 * g4 + i] += r;
              s2[0][4 * g4 + i] += r * r;
            }
            if (BS) {
              float gg = Elem<T>::to_f(vals[i]);       // the gradient as stored
              const float yc = Elem<T>::to_f(reinterpret_cast<const T*>(&yq[b][g4])[i]) - bmean[4 * g4 + i];
              if (!(yc > 0.f)) gg *= g.bs_slope;       // (slope == 1: no activation, nothing changes)
              s1[0][4 * g4 + i] += gg;
              s2[0][4 * g4 + i] += gg * yc;
            }
          }
          piece[g4] = *reinterpret_cast<u32x2*>(vals);
          if (!RX_ST16) *reinterpret_cast<u32x2*>(op + co) = piece[g4];
        }
        // 16-byte stores (rx_pair16; both lanes of a voxel pass the bounds test together): 2-5 % on this kernel.  The same change
        // measured -1..-2 % on conv_halo64ws and on the full-resolution conv_halo32 launch, so those keep their 8-byte stores
        if (RX_ST16) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            *reinterpret_cast<u32x4*>(op + 16 * pr + 8 * fh) = rx_pair16(piece[2 * pr], piece[2 * pr + 1]);
        }
      }
      lds_only_barrier();      // the stores of this tile stay in flight under the next tile's MFMAs
    }
    if (STATS || BS) ch_stat_flush<1>(s1, s2, g.stat_part, sn, g.wgs_s * 4, sl * 4 + wave, g.Co, 0, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised, persistent variant of conv_halo_kernel<T,64> on the 4x4x16 tile (the 64/128-channel layers at 64^3
// and 32^3 and the 64-channel data gradients at 128^3).  The generic kernel runs at 35 % MFMA / 36 % LDS utilisation: per
// (chunk, dz) phase a workgroup stops twice at a barrier, commits 37-78 KB to LDS and only then issues the next phase's
// loads.  Here 4 PRODUCER waves stream the next phase's weight plane (and, at a chunk boundary, the next halo) through
// registers into the other LDS buffer while 4 CONSUMER waves run the phase's 72 MFMAs each from the current one:
//   LDS = 2 x (9 x 64 weight rows + 648 halo rows) x 64 B = 156,672 B -> one workgroup per CU, one barrier per phase.
// Rows stay 64 bytes, XOR-swizzled on (row>>2)&3; voxels are dealt to lanes by lane_voxel() (conflict-free lane groups) and
// the swizzle term of (lane row + tap offset) comes out of a per-lane 32-bit table (one v_bfe per tap and voxel block).
// A workgroup walks a contiguous range of tiles for one 64-channel output block.
// ---------------------------------------------------------------------------------------------------------------------
#define CH64_W_BYTES (9 * 64 * 64)
#define CH64_X_BYTES (648 * 64)

// BSP (data-gradient launches that COMPLETE the output gradient of an InstanceNorm layer without residual -- see ConvHaloGeom::bs_y):
// the layer's two backward sums come out of this launch.  The consumer waves have no registers to spare for them (conv_halo32p keeps
// them there), so they go to the PRODUCER waves, which idle between DMA issues: a tile's output is staged in LDS as bf16 (the halo
// buffer of the tile's last chunk is free from the barrier that ends the tile until the DMA issued in the next tile's second phase;
// 144-byte voxel rows), the consumers read it back for whole-row 16-byte global stores, the producers read it once more, beside the
// layer's y (fetched during the tile's last phase), and keep running sums of g' and g' * (y - mean) for 8 channels per thread; one
// partial row per producer wave at the end, laid out like ch_stat_flush's.  One more barrier per tile; alone the staging is neutral
// (DESIGN 10.4 row bf).
// NA: 32-channel output blocks per workgroup (2: the 64-channel block this kernel was written for; 1: layers whose (tile, 64-channel
// block) pairs would leave CUs idle -- 256 channels at 16^3 -- run as (tile, 32-channel block) pairs: half the weight plane per phase,
// half the MFMAs, the same halo; LDS-DMA variants only).
template <typename T, bool FLIP, bool GLDS, bool XDMA = false, bool STATS = false, bool ACC = false, bool BSP = false, int NA = 2>
__global__ __launch_bounds__(512, 1) void conv_halo64ws_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                               T* __restrict__ out, const ConvHaloGeom g, int tiles_per_wg) {
  constexpr int P = Elem<T>::PER16;
  constexpr int KB = 4 * P;
  constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2, HV = 648;
  constexpr int XPIECES = (HV * 4 + 255) / 256;   // 11
  static_assert(NA == 2 || (GLDS && XDMA && !BSP), "32-channel blocks: LDS-DMA variants without backward sums only");
  constexpr int BNW = 32 * NA;                    // output channels per workgroup
  constexpr int W_BYTES = 9 * BNW * 64;           // one weight plane
  constexpr int WINSTR = 9 * BNW / 16;            // 1-KiB DMA pieces of a plane (18 / 9 per 4 producer waves)
  constexpr int WPIECES = (WINSTR + 3) / 4;       // 9 (NA = 2) / 5, the last one on two waves only (NA = 1)
  constexpr int CENTRE = HY * HX + HX + 1;        // 127
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sWb = smem;                        // [2][9*64][64 B]
  unsigned char* sXb = smem + 2 * W_BYTES;          // [2][648][64 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lid = g.order ? rx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) : blockIdx.x * gridDim.y + blockIdx.y;
  const int vb = lid / gridDim.y;
  int t_begin = vb * tiles_per_wg, t_end = min(g.NT, t_begin + tiles_per_wg);
  const int NTs = g.NT / g.N;                          // tiles per sample
  const int sn = g.wgs_s ? vb / g.wgs_s : 0, sl = g.wgs_s ? vb - sn * g.wgs_s : 0;
  if (g.wgs_s) t_begin = sn * NTs + sl * tiles_per_wg, t_end = min((sn + 1) * NTs, t_begin + tiles_per_wg);
  const int n0 = (lid - vb * gridDim.y) * BNW;
  const int nchunks = g.Ci / KB;
  const int ppt = 3 * nchunks;                      // phases per tile
  const int nphase = (t_end - t_begin) * ppt;

  auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
    int tx, ty, tz;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
    z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
  };

  if (wave >= 4) {
    // ================================= producers =================================
    const int ptid = tid - 256, chunk = ptid & 3;
    int xh[XPIECES];
#pragma unroll
    for (int p = 0; p < XPIECES; ++p) {
      const int row = (ptid >> 2) + 64 * p;
      const int hx = row % HX, t = row / HX;
      xh[p] = row < HV ? ((t / HY) << 16) | ((t % HY) << 8) | hx : -1;
    }
    int woff[WPIECES];
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      const int i = ptid + 256 * p;                  // piece ((tl*64 + r)*4 + c4)
      const int c4 = i & 3, r = (i >> 2) % BNW, tl = (i >> 2) / BNW;
      woff[p] = ((tl * g.Co + n0 + r) * g.Ci) + c4 * P;
    }
    const long wplane = (long)9 * g.Co * g.Ci;
    // GLDS: the weight plane goes global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write).  One
    // wave-instruction writes 1 KiB = 16 rows x 64 B lane-linearly, so the XOR swizzle is applied to the SOURCE address:
    // lane l of piece j holds row (j*4 + pw)*16 + (l>>2), data chunk (l&3) ^ ((row>>2)&3).
    const int pw = wave - 4, pl = lane;
    int goff[WPIECES];
#pragma unroll
    for (int p = 0; p < WPIECES; ++p) {
      const int rr = (p * 4 + pw) * 16 + (pl >> 2);        // tl*BNW + r
      const int tl = rr / BNW, r = rr % BNW, c4 = (pl & 3) ^ ((rr >> 2) & 3);
      goff[p] = ((tl * g.Co + n0 + r) * g.Ci) + c4 * P;
    }
    u32x4 xr[XPIECES], wr[WPIECES];
    auto dma_weights = [&](int ph) {
      if (RX_ABLATE(g, 2)) return;
      const int r = ph % ppt, cc = r / 3, dzg = r - cc * 3;
      const T* wp = w + dzg * wplane + cc * KB;
      __attribute__((address_space(3))) unsigned char* dst =
          (__attribute__((address_space(3))) unsigned char*)(sWb + (ph & 1) * W_BYTES);
#pragma unroll
      for (int p = 0; p < WPIECES; ++p)
        if (p * 4 + pw < WINSTR)              // (wave-uniform; the counted waits below only count the halo pieces issued after these)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + goff[p]),
                                         (__attribute__((address_space(3))) void*)(dst + (p * 4 + pw) * 1024), 16, 0, 0);
    };
    // XDMA: the halo goes global -> LDS by buffer_load ... lds as well (same lane-linear 1 KiB pieces, so the row swizzle
    // is again applied on the source side: (row>>2)&3 == (lane>>4)&3 for every piece); voxels outside the volume get an
    // out-of-range buffer offset, for which the hardware writes ZEROS (scripts/probes/buffer_lds_oob.hip); the lanes of
    // rows >= 648 in the last piece are switched off.  No staging registers, no ds_write pass.
    const int schunk = (pl & 3) ^ ((pl >> 4) & 3);
    __amdgpu_buffer_rsrc_t rX;
    if (XDMA) rX = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (unsigned)(((long)g.N * g.in_ss + (g.in_cs == KB ? 0L : (long)(g.Ci / KB - 1) * g.in_cs)) * 2), 0x00020000);   // (planar concat: plane j at j * in_cs)
    auto dma_halo = [&](int ph) {
      if (RX_ABLATE(g, 1)) return;
      const int tile = t_begin + ph / ppt, cc = (ph % ppt) / 3;
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      const unsigned base = (unsigned)(((long)n * g.in_ss + cc * g.in_cs + schunk * P) * 2);     // (in_cs == KB unless the input is a planar concat)
      __attribute__((address_space(3))) unsigned char* dst =
          (__attribute__((address_space(3))) unsigned char*)(sXb + ((ph / 3) & 1) * CH64_X_BYTES);
#pragma unroll
      for (int p = 0; p < XPIECES; ++p) {
        if (xh[p] >= 0) {
          const int z = z0 + (xh[p] >> 16) - 1, y = y0 + ((xh[p] >> 8) & 255) - 1, x = x0 + (xh[p] & 255) - 1;
          const bool ok = (unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X;
          const unsigned off = ok ? base + (unsigned)(((z * g.Y + y) * g.X + x) * g.ldi * 2) : 0x7fffff00u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)(dst + (p * 4 + pw) * 1024), 16, off, 0, 0, 0);
        }
      }
    };
    auto load_phase = [&](int ph) {                  // ph relative to this workgroup
      const int tile = t_begin + ph / ppt, r = ph % ppt, cc = r / 3, dzg = r - cc * 3;
      const T* wp = w + dzg * wplane + cc * KB;
      if (!GLDS) {
#pragma unroll
        for (int p = 0; p < WPIECES; ++p) wr[p] = *reinterpret_cast<const u32x4*>(wp + woff[p]);
      }
      if (dzg == 0) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        const T* in_n = in + (long)n * g.in_ss + cc * g.in_cs + chunk * P;
#pragma unroll
        for (int p = 0; p < XPIECES; ++p) {
          u32x4 v = u32x4{0u, 0u, 0u, 0u};
          if (xh[p] >= 0) {
            const int z = z0 + (xh[p] >> 16) - 1, y = y0 + ((xh[p] >> 8) & 255) - 1, x = x0 + (xh[p] & 255) - 1;
            if ((unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X)
              v = *reinterpret_cast<const u32x4*>(in_n + ((long)(z * g.Y + y) * g.X + x) * g.ldi);
          }
          xr[p] = v;
        }
      }
    };
    auto commit_phase = [&](int ph) {
      unsigned char* sW = sWb + (ph & 1) * W_BYTES;
      if (!GLDS) {
#pragma unroll
        for (int p = 0; p < WPIECES; ++p) {
          const int i = ptid + 256 * p;
          const int c4 = i & 3, rr = i >> 2;           // rr = tl*64 + r
          *reinterpret_cast<u32x4*>(sW + rr * 64 + ((c4 ^ ((rr >> 2) & 3)) << 4)) = wr[p];
        }
      }
      if (ph % 3 == 0) {                              // (ppt is a multiple of 3: dz == 0 <=> ph % 3 == 0)
        unsigned char* sX = sXb + ((ph / 3) & 1) * CH64_X_BYTES;
#pragma unroll
        for (int p = 0; p < XPIECES; ++p) {
          const int row = (ptid >> 2) + 64 * p;
          if (xh[p] >= 0) *reinterpret_cast<u32x4*>(sX + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)) = xr[p];
        }
      }
    };
    if (XDMA) {
      // ---- BSP: this thread's share of the backward sums: 16-byte chunk bc (8 channels) of voxels bvg + 32 p of every tile
      const int bc = ptid & 7, bvg = ptid >> 3;
      float bs1[8], bs2[8], bmean[8];
      u32x4 yv[8];
      if (BSP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bs1[j] = 0.f, bs2[j] = 0.f, bmean[j] = g.bs_stats[2 * ((size_t)sn * g.Co + n0 + 8 * bc + j)];
      }
      auto bs_load_y = [&](int tile) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        const T* yp = (const T*)g.bs_y + (long)n * g.bs_yss + n0 + 8 * bc;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int v = bvg + 32 * p;
          const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
          yv[p] = (z < g.Z && y < g.Y && x < g.X) ? *reinterpret_cast<const u32x4*>(yp + ((long)(z * g.Y + y) * g.X + x) * g.bs_ldy) : u32x4{0u, 0u, 0u, 0u};
        }
      };
      auto bs_accumulate = [&](int tile, const unsigned char* sS) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int v = bvg + 32 * p;
          const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
          if (z < g.Z && y < g.Y && x < g.X) {
            const u32x4 gq = *reinterpret_cast<const u32x4*>(sS + v * 144 + bc * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float gg = Elem<T>::to_f(reinterpret_cast<const T*>(&gq)[j]);          // the gradient as stored
              const float yc = Elem<T>::to_f(reinterpret_cast<const T*>(&yv[p])[j]) - bmean[j];
              if (!(yc > 0.f)) gg *= g.bs_slope;
              bs1[j] += gg;
              bs2[j] += gg * yc;
            }
          }
        }
      };
      if (nphase > 0) {
        dma_weights(0);
        dma_halo(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_only_barrier();
      int pit = 0, pending = -1;             // pending: tile whose staged output waits for bs_accumulate
      const unsigned char* pendS = nullptr;
      for (int ph = 0; ph < nphase; ++ph) {
        const bool tile_end = pit == ppt - 1;
        if (ph + 1 < nphase) {
          dma_weights(ph + 1);
          if (ph + 2 < nphase && (ph + 2) % 3 == 0) {
            // halo of the next chunk into the buffer last read two chunks ago; it only has to land before the barrier
            // that ends iteration ph+1, so it stays in flight across this one: wave 0 of the producers issues 11 pieces
            // (the last one partly masked), the others 10 (their 11th is entirely beyond row 648)
            dma_halo(ph + 2);
            if (pw == 0)
              asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else
              asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
          } else {
            // (a tile's first phase, behind this phase's DMA issue: the previous tile's staged output is still in its halo buffer --
            // the DMA that overwrites it is issued in the NEXT iteration; a tile's last phase: fetch its y for the sums)
            if (BSP && pending >= 0) {
              bs_accumulate(pending, pendS);
              pending = -1;
            }
            if (BSP && tile_end) bs_load_y(t_begin + ph / ppt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        } else if (BSP) {                      // the workgroup's very last phase
          bs_load_y(t_begin + ph / ppt);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_only_barrier();
        if (++pit == ppt) {
          pit = 0;
          if (BSP) {                           // the consumers stage the tile between this barrier and the next one
            lds_only_barrier();
            pending = t_begin + ph / ppt;
            pendS = sXb + ((ph / 3) & 1) * CH64_X_BYTES;
          }
        }
      }
      if (BSP) {
        if (pending >= 0) bs_accumulate(pending, pendS);
        // lanes 8 vg + bc of a wave hold the same 8 channels: fold the 8 voxel groups, one partial row per producer wave
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float u = bs1[j], v = bs2[j];
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) {
            u += __shfl_xor(u, o, 64);
            v += __shfl_xor(v, o, 64);
          }
          bs1[j] = u, bs2[j] = v;
        }
        if (pl < 8) {
          float* p0 = g.stat_part + ((size_t)(sn * (g.wgs_s * 4) + sl * 4 + pw) * 2) * g.Co + n0 + 8 * pl;
#pragma unroll
          for (int j = 0; j < 8; ++j) p0[j] = bs1[j], p0[g.Co + j] = bs2[j];
        }
      }
    } else {
    if (nphase > 0) {
      if (GLDS) dma_weights(0);
      load_phase(0);
      commit_phase(0);
      if (nphase > 1) load_phase(1);
      if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_only_barrier();
    for (int ph = 0; ph < nphase; ++ph) {
      if (ph + 1 < nphase) {
        if (GLDS) dma_weights(ph + 1);                // in flight behind the register-staged work of this iteration
        commit_phase(ph + 1);                         // buffers of phase ph+1 were last read in phase ph-1 / chunk-2
        if (ph + 2 < nphase) load_phase(ph + 2);
        if (GLDS) {                                   // the DMAs must have landed; the halo loads issued after them may fly on
          if (ph + 2 < nphase && (ph + 2) % 3 == 0)
            asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      lds_only_barrier();
    }
    }
  } else {
    // ================================= consumers =================================
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const int fr = lane & 31, fh = lane >> 5;
    const int fv = lane_voxel(fr);
    unsigned xorg[2], swt[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int v = (wave * 2 + b) * 32 + fv;
      const int vx = v & 15, vy = (v >> 4) & 3, vz = v >> 6;
      const int corner = (vz * HY + vy) * HX + vx;      // row of tap (-1,-1,-1) of this voxel
      xorg[b] = corner * 64;
      unsigned t = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t |= (unsigned)(fh ^ (((corner + k) >> 2) & 3)) << (2 * k);
      swt[b] = t;
    }
    const int wsw = (fr >> 2) & 3;
    const unsigned wl0 = fr * 64 + (((0 + fh) ^ wsw) << 4);   // k-step 0; rows tl*64 + a*32 keep (row>>2)&3
    const unsigned wl1 = fr * 64 + (((2 + fh) ^ wsw) << 4);   // k-step 1
    const unsigned sW_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)sWb;
    const unsigned sX_base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)sXb;

    f32x16 acc[NA][2];
    auto plane = [&](auto dzc, unsigned wbase, unsigned xbase) {
      constexpr int DZG = decltype(dzc)::value;
      constexpr int DEPTH = STATS ? 2 : 3;      // the 64 statistics accumulators leave room for a 3-slot fragment ring only
      u32x4 fa[DEPTH + 1][NA], fb[DEPTH + 1][2];
      unsigned xa[2];
      auto ld = [&](int i, int slot) {
        const int tl = i >> 1, ks = i & 1;
        if (ks == 0) {
          const int dz = FLIP ? 1 - DZG : DZG - 1, dy = FLIP ? 1 - tl / 3 : tl / 3 - 1, dx = FLIP ? 1 - tl % 3 : tl % 3 - 1;
          const int K = CENTRE + (dz * HY + dy) * HX + dx;
#pragma unroll
          for (int b = 0; b < 2; ++b) xa[b] = xbase + xorg[b] + K * 64 + (((swt[b] >> (2 * (K & 15))) & 3u) << 4);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[slot][a] = *(const lds_u32x4*)(uintptr_t)(wbase + (ks ? wl1 : wl0) + (tl * BNW + a * 32) * 64);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[slot][b] = *(const lds_u32x4*)(uintptr_t)(ks ? xa[b] ^ 32u : xa[b]);
      };
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) ld(i, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        if (i + DEPTH < 18) ld(i + DEPTH, (i + DEPTH) % (DEPTH + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) Mma<T>::run(acc[a][b], fa[i % (DEPTH + 1)][a], fb[i % (DEPTH + 1)][b]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    float s1[NA][16], s2[NA][16];
    if (STATS) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[a][r] = 0.f, s2[a][r] = 0.f;
    }
    lds_only_barrier();
    int ph = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
      u32x2 oldv[NA][2][4];                               // accumulate: old dx values fetched under the MFMA loops (see conv_halo32p)
      if (ACC) {
        int n_, z0_, y0_, x0_;
        tile_origin(tile, n_, z0_, y0_, x0_);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int v = (wave * 2 + b) * 32 + fv;
          const int z = z0_ + (v >> 6), y = y0_ + ((v >> 4) & 3), x = x0_ + (v & 15);
          const bool ok = z < g.Z && y < g.Y && x < g.X;
          const T* op = out + (long)n_ * g.out_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldo + n0 + 4 * fh;
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) oldv[a][b][g4] = ok ? *reinterpret_cast<const u32x2*>(op + a * g.out_cs + 8 * g4) : u32x2{0u, 0u};
        }
      }
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      for (int cc = 0; cc < nchunks; ++cc) {
        const unsigned xbase = sX_base + ((ph / 3) & 1) * CH64_X_BYTES;
        if (!RX_ABLATE(g, 4)) plane(std::integral_constant<int, 0>{}, sW_base + (ph & 1) * W_BYTES, xbase);
        lds_only_barrier();
        ++ph;
        if (!RX_ABLATE(g, 4)) plane(std::integral_constant<int, 1>{}, sW_base + (ph & 1) * W_BYTES, xbase);
        lds_only_barrier();
        ++ph;
        if (!RX_ABLATE(g, 4)) plane(std::integral_constant<int, 2>{}, sW_base + (ph & 1) * W_BYTES, xbase);
        if (cc + 1 < nchunks) {
          lds_only_barrier();
          ++ph;
        }
      }
      // ---- epilogue of this tile (its stores stay in flight), then the barrier that ends the tile's last phase.  BSP: the tile
      // goes through LDS (see the kernel's header): barrier, stage, barrier, whole-row stores
      int n, z0, y0, x0;
      tile_origin(tile, n, z0, y0, x0);
      constexpr int SROW = 144;
      unsigned char* sS = sXb + ((ph / 3) & 1) * CH64_X_BYTES;
      if (BSP) lds_only_barrier();
      int fv2 = fv;
      if (BSP) asm volatile("" : "+v"(fv2));      // (recomputed per tile: hoisted out of the tile loop the staging addresses cost spills)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int v = (wave * 2 + b) * 32 + fv2;
        const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
        if (z >= g.Z || y >= g.Y || x >= g.X || RX_ABLATE(g, 8)) continue;
        T* op = out + (long)n * g.out_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldo + n0;
        auto epi = [&](auto HB, auto RA) {
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int co = a * 32 + 8 * g4 + 4 * fh;
              const long cm = a * g.out_cs + 8 * g4 + 4 * fh;    // memory offset of channel co (== co unless out is a planar concat)
              f32x4 bv = {0.f, 0.f, 0.f, 0.f};
              if (decltype(HB)::value) bv = *reinterpret_cast<const f32x4*>(bias + n0 + co);
              u32x2 old2 = {0u, 0u};
              if (!ACC && decltype(RA)::value) old2 = *reinterpret_cast<const u32x2*>(op + cm);
              T vals[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float f = acc[a][b][4 * g4 + i] + bv[i];
                if (ACC) f += Elem<T>::to_f(reinterpret_cast<const T*>(&oldv[a][b][g4])[i]);
                else if (decltype(RA)::value) f += Elem<T>::to_f(reinterpret_cast<const T*>(&old2)[i]);
                vals[i] = Elem<T>::from_f(f);
                if (STATS) {
                  const float r = Elem<T>::to_f(vals[i]);
                  s1[a][4 * g4 + i] += r;
                  s2[a][4 * g4 + i] += r * r;
                }
              }
              if (BSP) *reinterpret_cast<u32x2*>(sS + v * SROW + co * 2) = *reinterpret_cast<u32x2*>(vals);
              else if (!RX_ABLATE(g, 16) || vals[0] == (T)12345.f) *reinterpret_cast<u32x2*>(op + cm) = *reinterpret_cast<u32x2*>(vals);
            }
        };
        RX_EPI_DISPATCH(bias != nullptr, !ACC && g.accumulate != 0, epi);
      }
      lds_only_barrier();
      if (BSP) {
        // piece i = tid + 256 p of the tile: voxel (tid >> 3) + 32 p = (z p>>1, y (tid>>7) + 2 (p&1), x (tid>>3)&15), 16-byte chunk tid & 7
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int ry = t2 >> 7, rx = (t2 >> 3) & 15, c = t2 & 7;
        T* gp = out + (long)n * g.out_ss + ((long)(z0 * g.Y + y0 + ry) * g.X + x0 + rx) * g.ldo + n0 + (c >> 2) * g.out_cs + (c & 3) * 8;
        const unsigned char* lp = sS + (t2 >> 3) * SROW + c * 16;
        const bool okx = x0 + rx < g.X;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int zz = p >> 1, yy = 2 * (p & 1);
          if (okx && z0 + zz < g.Z && y0 + ry + yy < g.Y)
            *reinterpret_cast<u32x4*>(gp + ((long)(zz * g.Y + yy) * g.X) * g.ldo) = *reinterpret_cast<const u32x4*>(lp + p * 32 * SROW);
        }
      }
      ++ph;
    }
    if (STATS) ch_stat_flush<NA>(s1, s2, g.stat_part, sn, g.wgs_s * 4, sl * 4 + wave, g.Co, n0, lane);
  }
}

template <typename T>
static void ch64ws_launch(hipStream_t st, const void* in, const void* w, const float* bias, void* out, const ConvHaloGeom& g) {
  const size_t lds = (size_t)2 * (CH64_W_BYTES + CH64_X_BYTES);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int cob = g.Co / 64;
  int wgs = 256 / cob;                                   // one persistent workgroup per CU in total
  if (wgs < 1) wgs = 1;
  if (wgs > g.NT) wgs = g.NT;
  // a workgroup's tile range never straddles samples (the fused InstanceNorm statistics are per (n, c))
  const int NTs = g.NT / g.N;
  int wgs_s = wgs / g.N > 0 ? wgs / g.N : 1;
  const int per = (NTs + wgs_s - 1) / wgs_s;
  wgs_s = (NTs + per - 1) / per;
  wgs = wgs_s * g.N;
  const_cast<ConvHaloGeom&>(g).wgs_s = wgs_s;
  dim3 grid(wgs, cob);
  static int glds = -1;
  if (glds < 0) {
    const char* e = getenv("RX_CH64_GLDS");
    glds = e ? atoi(e) : 1;   // default: weight planes by LDS-DMA (+2-4 % isolated, -0.1 ms per step, bit-identical)
  }
  static int xdma = -1;
  if (xdma < 0) {
    const char* e = getenv("RX_CH64_XDMA");
    xdma = e ? atoi(e) : 1;   // default: the halo by LDS-DMA too
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const bool dma_ok = glds && xdma && (long)g.N * g.in_ss * 2 < 0x7fffff00L;   // (out-of-range offsets must stay out of range of the descriptor)
  if (dma_ok && g.stat_part && !g.flip && !g.bs_y) {
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_s = true;
    }
    hipLaunchKernelGGL((conv_halo64ws_kernel<T, false, true, true, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    return;
  }
  if (dma_ok && g.bs_y && g.stat_part && g.flip && !((uintptr_t)out & 15) && g.ldo % 8 == 0 && g.out_cs % 8 == 0 && !((uintptr_t)g.bs_y & 15) &&
      g.bs_ldy % 8 == 0) {     // backward sums on the producer waves (whole-row 16-byte stores / loads)
    static bool attr_b = false;
    if (!attr_b) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_b = true;
    }
    if (g.accumulate)
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, true, true, false, true, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    else
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, true, true, false, false, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    return;
  }
  const_cast<ConvHaloGeom&>(g).bs_y = nullptr;                   // (not taken: the caller runs the separate reduce pass)
  const_cast<ConvHaloGeom&>(g).stat_part = nullptr;              // only the instantiations above accumulate statistics
  if (glds && xdma && g.accumulate && g.flip && (long)g.N * g.in_ss * 2 < 0x7fffff00L) {     // dx +=: old values prefetched
    static bool attr_a = false;
    if (!attr_a) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_a = true;
    }
    hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, true, true, false, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    return;
  }
  if (glds && xdma && (long)g.N * g.in_ss * 2 < 0x7fffff00L) {   // out-of-range offsets must stay out of range of the descriptor
    if (g.flip)
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, true, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    else
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, false, true, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
  } else if (glds) {
    if (g.flip)
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
    else
      hipLaunchKernelGGL((conv_halo64ws_kernel<T, false, true>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
  } else if (g.flip)
    hipLaunchKernelGGL((conv_halo64ws_kernel<T, true, false>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
  else
    hipLaunchKernelGGL((conv_halo64ws_kernel<T, false, false>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per);
}

// the 32-channel-block instantiations (NA = 1) of conv_halo64ws: LDS-DMA only.  Returns false (nothing launched) when the DMA
// preconditions do not hold -- the caller then takes conv_halo32_kernel as before.
template <typename T>
static bool ch32ws_launch(hipStream_t st, const void* in, const void* w, const float* bias, void* out, const ConvHaloGeom& g) {
  const size_t lds = (size_t)2 * (9 * 32 * 64 + CH64_X_BYTES);
  if (!(((long)g.N * g.in_ss + (g.in_cs == 32 ? 0L : (long)(g.Ci / 32 - 1) * g.in_cs)) * 2 < 0x7fffff00L)) return false;   // (planar concat input: the planes lie in_cs apart)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, true, true, false, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, false, true, true, true, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true, false, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo64ws_kernel<T, true, true, true, false, true, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int cob = g.Co / 32;
  int wgs = 256 / cob;                                   // one persistent workgroup per CU in total
  if (wgs < 1) wgs = 1;
  if (wgs > g.NT) wgs = g.NT;
  const int NTs = g.NT / g.N;                            // a workgroup's tile range never straddles samples
  int wgs_s = wgs / g.N > 0 ? wgs / g.N : 1;
  const int per = (NTs + wgs_s - 1) / wgs_s;
  wgs_s = (NTs + per - 1) / per;
  wgs = wgs_s * g.N;
  const_cast<ConvHaloGeom&>(g).wgs_s = wgs_s;
  dim3 grid(wgs, cob);
#define RX_32WS(F, S, A) hipLaunchKernelGGL((conv_halo64ws_kernel<T, F, true, true, S, A, false, 1>), grid, dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per)
  if (g.stat_part && !g.flip && !g.accumulate) {
    RX_32WS(false, true, false);
    return true;
  }
  const_cast<ConvHaloGeom&>(g).stat_part = nullptr;
  if (g.flip && g.accumulate) RX_32WS(true, false, true);
  else if (g.flip) RX_32WS(true, false, false);
  else RX_32WS(false, false, false);
#undef RX_32WS
  return true;
}

template <typename T>
static void ch32p_launch(hipStream_t st, const void* in, const void* w, const float* bias, void* out, const ConvHaloGeom& g) {
  const size_t lds = (size_t)CH32P_W_BYTES + 2 * (size_t)CH32P_HALO_BYTES;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32p_kernel<T, false, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  int wgs = g.NT < 256 ? g.NT : 256;                    // one persistent workgroup per CU
  const int NTs = g.NT / g.N;                           // a workgroup's tile range never straddles samples
  int wgs_s = wgs / g.N > 0 ? wgs / g.N : 1;
  const int per = (NTs + wgs_s - 1) / wgs_s;
  wgs_s = (NTs + per - 1) / per;
  wgs = wgs_s * g.N;
  const_cast<ConvHaloGeom&>(g).wgs_s = wgs_s;
#define RX_32P(S, A, F) hipLaunchKernelGGL((conv_halo32p_kernel<T, S, A, F>), dim3(wgs), dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per)
#define RX_32PB(A) hipLaunchKernelGGL((conv_halo32p_kernel<T, false, A, true, true>), dim3(wgs), dim3(512), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, per)
  if (g.bs_y && g.flip && g.accumulate) RX_32PB(true);
  else if (g.bs_y && g.flip) RX_32PB(false);
  else
#undef RX_32PB
  if (g.stat_part && !g.accumulate && !g.flip) RX_32P(true, false, false);
  else if (g.flip && g.accumulate) RX_32P(false, true, true);
  else if (g.flip) RX_32P(false, false, true);
  else if (g.accumulate) RX_32P(false, true, false);
  else RX_32P(false, false, false);
#undef RX_32P
}

template <typename T>
static void ch32_launch(dim3 grid, hipStream_t st, const void* in, const void* w, const float* bias, void* out, const ConvHaloGeom& g) {
  const size_t lds = (size_t)648 * 80 + (size_t)9 * 32 * 64;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (g.stat_part && !g.flip && !g.accumulate) {
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo32_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_s = true;
    }
    hipLaunchKernelGGL((conv_halo32_kernel<T, false, true>), grid, dim3(256), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g);
  } else if (g.flip)
    hipLaunchKernelGGL((conv_halo32_kernel<T, true>), grid, dim3(256), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g);
  else
    hipLaunchKernelGGL((conv_halo32_kernel<T, false>), grid, dim3(256), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g);
}

static int ch_p2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static int ch_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <typename T>
static void ch_dispatch(int BN, dim3 grid, hipStream_t st, const void* in, const void* w, const float* bias, void* out,
                        const ConvHaloGeom& g, float* slab = nullptr, int cps = 0) {
  if (BN == 64) {
    const size_t lds = (size_t)(RX_CH_MAX_HV * 4 + 9 * 64 * 4) * 16;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<T, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<T, 64>), grid, dim3(256), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, slab, cps);
  } else {
    const size_t lds = (size_t)(RX_CH_MAX_HV * 4 + 9 * 32 * 4) * 16;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<T, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<T, 32>), grid, dim3(256), lds, st, (const T*)in, (const T*)w, bias, (T*)out, g, slab, cps);
  }
}

// returns 1 if handled, 0 to fall through to the generic kernel, negative on error.
// in/out: same spatial dims (stride 1, kernel 3x3x3, padding 1).  flip = 1 for backward-data.
// stat_part / stat_chunks (forward only, optional): when the launch goes to a persistent kernel, per-wave partial sums of
// y and y^2 are left in stat_part ([n][*stat_chunks][2][Co] floats) and *stat_chunks > 0; otherwise *stat_chunks = 0.
int rx_conv_halo_try(rx_dtype dt, const rx_act* in, const void* w, const float* bias, const rx_act* out, int flip, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st, float* stat_part, size_t stat_bytes, int* stat_chunks,
                     const RxBwdStat* bs) {
  if (stat_chunks) *stat_chunks = 0;
  const int per16 = dt == RX_F32 ? 4 : 8;
  const int KB = 4 * per16;
  if (in->c % KB || out->c % 32 || in->ld % per16 || out->ld % 4) return 0;
  if (((uintptr_t)in->ptr & 15) || ((uintptr_t)out->ptr & 7) || ((uintptr_t)w & 15)) return 0;
  if (rx_act_voxels(in) * (long)in->ld >= (1L << 31)) return 0;  // 32-bit halo offsets
  ConvHaloGeom g;
  memset(&g, 0, sizeof(g));
  g.N = in->n, g.Z = in->z, g.Y = in->y, g.X = in->x;
  g.Ci = in->c, g.Co = out->c, g.ldi = in->ld, g.ldo = out->ld;
  g.in_ss = rx_act_voxels(in) * (long)in->ld;
  g.out_ss = rx_act_voxels(out) * (long)out->ld;
  g.in_cs = in->cs ? in->cs : KB, g.out_cs = out->cs ? out->cs : 32;
  if ((in->cs || out->cs) && dt == RX_F32) return 0;
  int TX = ch_p2ceil(g.X);
  TX = TX < 4 ? 4 : (TX > 16 ? 16 : TX);
  int rem = 256 / TX;
  int TY = ch_p2ceil(g.Y);
  int capy = TX == 16 ? 4 : 8;
  if (TY > capy) TY = capy;
  if (TY > rem) TY = rem;
  int TZ = ch_p2ceil(g.Z);
  if (TZ > rem / TY) TZ = rem / TY;
  if (TZ * TY * TX != 256) return 0;  // the kernel is written for full 256-voxel tiles
  g.TZ = TZ, g.TY = TY, g.TX = TX, g.lTX = ch_ilog2(TX), g.lTY = ch_ilog2(TY);
  g.VT = 256;
  g.HY = TY + 2, g.HX = TX + 2, g.HV = (TZ + 2) * g.HY * g.HX;
  if (g.HV > RX_CH_MAX_HV) return 0;
  g.tz_n = (g.Z + TZ - 1) / TZ, g.ty_n = (g.Y + TY - 1) / TY, g.tx_n = (g.X + TX - 1) / TX;
  g.NT = g.N * g.tz_n * g.ty_n * g.tx_n;
  g.accumulate = accumulate, g.flip = flip;
  {
    static int order = -1;   // RX_TILE_ORDER: 0 raster, 1 (default) locality-aware walk (rx_tile_coords), bit-identical results
    if (order < 0) {
      const char* e = getenv("RX_TILE_ORDER");
      order = e ? atoi(e) : 1;
    }
    g.order = order;         // persistent kernels: z-fastest columns; one-tile grids: 4x4x4 bricks where the tile grid allows
  }
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("RX_DBG");
      dbg = e ? atoi(e) : 0;
    }
    g.dbg = dbg;
  }
  int BN = (g.Co % 64 == 0) ? 64 : 32;
  if (BN == 64 && (long)g.NT * (g.Co / 64) < 256) BN = 32;  // under-filled grid: twice the workgroups, half the work each
  if ((in->cs || out->cs) && (long)g.NT * (g.Co / BN) < 192) RX_FAIL(RX_EUNSUPPORTED, "conv_halo: planar-concat operand on a layer too small for the halo kernels");
  if ((long)g.NT * (g.Co / BN) < 192) {     // (128 pairs -- the 1024-channel data gradient of the first decoder conv at 8^3 -- measured 93 us here, ~60 us on igemm_fat)
    // too few (tile, channel block) pairs to fill 256 CUs: split the input channels over blockIdx.z, fp32 slabs + a
    // fixed-order reduce.  Against the split-K gather kernel (igemm_fat) the activations are staged ONCE per chunk with
    // their halo instead of once per tap -- that kernel moves 340 MB through L2 for a 512->512 layer at 8^3 (216 MB of
    // it re-gathered activations).  MEASURED (rocprofv3, inside the cfg2 step): 35.7 us + 5.5 us reduce against 36 us +
    // 5 us for igemm_fat -- no gain: with 3-6 phases per workgroup the kernel is a chain of cold weight-slice fetches
    // (64-byte pieces of 1 KB rows), not an L2-bandwidth problem.  Kept behind RX_CH_SPLITK=1 (off by default).
    static int splitk = -1;
    if (splitk < 0) {
      const char* e = getenv("RX_CH_SPLITK");
      splitk = e ? atoi(e) : 0;
    }
    const int nchunks = g.Ci / KB;
    const long base = (long)g.NT * (g.Co / 64);      // the split path always runs the 64-channel-block kernel
    if (!splitk || !ws || g.Co % 64 || nchunks < 4 || base < 8) return 0;
    static int target = -1, minch = -1;
    if (target < 0) {
      const char* e = getenv("RX_CH_SPLITK_WGS");
      target = e ? atoi(e) : 256;
      const char* e2 = getenv("RX_CH_SPLITK_MINCH");
      minch = e2 ? atoi(e2) : 2;
    }
    int S = (int)((target + base - 1) / base);
    if (S > nchunks / minch) S = nchunks / minch;    // >= 2 chunks (6 phases) per workgroup
    if (S > 16) S = 16;
    const int cps = (nchunks + S - 1) / S;
    S = (nchunks + cps - 1) / cps;
    const long rows = (long)g.N * g.Z * g.Y * g.X;
    if (S < 2 || (size_t)S * rows * g.Co * sizeof(float) > ws_bytes) return 0;
    ConvHaloGeom gs = g;
    gs.order = 0;
    dim3 grid3(g.NT, g.Co / 64, S);
    rx_note_kernel("conv_halo_kernel<64,splitk>");
    const long total = rows * (g.Co / 4);
    const int G = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    switch (dt) {
      case RX_F32:
        ch_dispatch<float>(64, grid3, st, in->ptr, w, nullptr, out->ptr, gs, (float*)ws, cps);
        hipLaunchKernelGGL((ch_splitk_reduce<float>), dim3(G), dim3(256), 0, st, (const float*)ws, S, rows, g.Co, bias, (float*)out->ptr, g.ldo, accumulate);
        break;
      case RX_BF16:
        ch_dispatch<bf16_t>(64, grid3, st, in->ptr, w, nullptr, out->ptr, gs, (float*)ws, cps);
        hipLaunchKernelGGL((ch_splitk_reduce<bf16_t>), dim3(G), dim3(256), 0, st, (const float*)ws, S, rows, g.Co, bias, (bf16_t*)out->ptr, g.ldo, accumulate);
        break;
      case RX_F16:
        ch_dispatch<f16_t>(64, grid3, st, in->ptr, w, nullptr, out->ptr, gs, (float*)ws, cps);
        hipLaunchKernelGGL((ch_splitk_reduce<f16_t>), dim3(G), dim3(256), 0, st, (const float*)ws, S, rows, g.Co, bias, (f16_t*)out->ptr, g.ldo, accumulate);
        break;
      default: return 0;
    }
    hipError_t e6 = hipGetLastError();
    if (e6 != hipSuccess) {
      rx_set_error("conv_halo split-K: %s", hipGetErrorString(e6));
      return RX_ELAUNCH;
    }
    return 1;
  }
  dim3 grid(g.NT, g.Co / BN);
  if (BN == 32 && TZ == 4 && TY == 4 && TX == 16 && dt != RX_F32 && g.Ci == 32 && g.Co == 32 && g.NT >= 512 && !getenv("RX_NO_CH32P")) {
    rx_note_kernel("conv_halo32p_kernel");               // 32 -> 32 channels: persistent, weights stationary in LDS
    if (in->cs || out->cs) RX_FAIL(RX_EUNSUPPORTED, "conv_halo32p: planar-concat operand");
    const bool room = stat_part && stat_chunks && (size_t)g.N * 1024 * 2 * g.Co * sizeof(float) <= stat_bytes;
    const bool fuse32 = room && ((!flip && !accumulate && !bs) || (flip && bs));
    g.stat_part = fuse32 ? stat_part : nullptr;
    if (fuse32 && bs) {
      g.bs_y = bs->y->ptr, g.bs_ldy = bs->y->ld, g.bs_yss = rx_act_voxels(bs->y) * (long)bs->y->ld;
      g.bs_stats = bs->stats, g.bs_slope = bs->slope;
    }
    if (dt == RX_BF16)
      ch32p_launch<bf16_t>(st, in->ptr, w, bias, out->ptr, g);
    else
      ch32p_launch<f16_t>(st, in->ptr, w, bias, out->ptr, g);
    hipError_t e4 = hipGetLastError();
    if (e4 != hipSuccess) {
      rx_set_error("conv_halo32p: %s", hipGetErrorString(e4));
      return RX_ELAUNCH;
    }
    if (fuse32) *stat_chunks = g.wgs_s * 4;
    g.stat_part = nullptr, g.bs_y = nullptr;
    return 1;
  }
  ConvHaloGeom g1 = g;                                // one tile per workgroup: bricks of tiles where the tile grid allows
  if (g.order && g.tx_n % 4 == 0 && g.ty_n % 4 == 0 && g.tz_n % 4 == 0) g1.order = 2;
  if (BN == 32 && TZ == 4 && TY == 4 && TX == 16 && dt != RX_F32 && !out->cs && g.Ci >= 64 && (long)g.NT * (g.Co / 32) >= 256 &&
      !(accumulate && !flip)) {
    // 32-channel blocks on the wave-specialised DMA pipeline (the 256-channel layers at 16^3: 32 tiles x 8 blocks = one workgroup
    // per CU; alone 43 -> ~30 us against conv_halo32_kernel's one tile per workgroup)
    static int ws32 = -1;     // RX_CH32WS=0: conv_halo32_kernel
    if (ws32 < 0) {
      const char* e = getenv("RX_CH32WS");
      ws32 = e ? atoi(e) : 1;
    }
    if (ws32) {
      const bool fuse = stat_part && stat_chunks && !flip && !accumulate && (size_t)g.N * 1024 * 2 * g.Co * sizeof(float) <= stat_bytes;
      ConvHaloGeom g2 = g;
      g2.stat_part = fuse ? stat_part : nullptr;
      g2.bs_y = nullptr;
      const bool launched = dt == RX_BF16 ? ch32ws_launch<bf16_t>(st, in->ptr, w, bias, out->ptr, g2) : ch32ws_launch<f16_t>(st, in->ptr, w, bias, out->ptr, g2);
      if (launched) {
        rx_note_kernel("conv_halo32ws_kernel");
        hipError_t e7 = hipGetLastError();
        if (e7 != hipSuccess) {
          rx_set_error("conv_halo32ws: %s", hipGetErrorString(e7));
          return RX_ELAUNCH;
        }
        if (fuse && g2.stat_part) *stat_chunks = g2.wgs_s * 4;
        return 1;
      }
    }
  }
  if (BN == 32 && TZ == 4 && TY == 4 && TX == 16) {  // full-resolution layers: compile-time tile, padded rows
    if (out->cs) RX_FAIL(RX_EUNSUPPORTED, "conv_halo32: planar-concat output");
    rx_note_kernel("conv_halo32_kernel");
    // statistics in the epilogue only for DEEP layers (>= 4 input-channel chunks: >= 432 MFMAs per wave behind the one wavefront
    // reduction) with few tiles per sample (the finalize reads 4 partial rows per tile): the 16^3 layers
    const int NTs32 = g.NT / g.N;
    const bool fuse32n = stat_part && stat_chunks && !flip && !accumulate && dt != RX_F32 && g.Ci / KB >= 4 && NTs32 * 4 <= 1024 &&
                         (size_t)g.N * NTs32 * 4 * 2 * g.Co * sizeof(float) <= stat_bytes;
    g1.stat_part = fuse32n ? stat_part : nullptr;
    switch (dt) {
      case RX_F32: ch32_launch<float>(grid, st, in->ptr, w, bias, out->ptr, g1); break;
      case RX_BF16: ch32_launch<bf16_t>(grid, st, in->ptr, w, bias, out->ptr, g1); break;
      case RX_F16: ch32_launch<f16_t>(grid, st, in->ptr, w, bias, out->ptr, g1); break;
      default: return 0;
    }
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) {
      rx_set_error("conv_halo32: %s", hipGetErrorString(e2));
      return RX_ELAUNCH;
    }
    if (fuse32n) *stat_chunks = NTs32 * 4;
    return 1;
  }
  static int ch64ws = -1;   // RX_CH64WS: 0 off, 1 (default) on for layers with >= 256 (tile, channel block) pairs, 2 always
  if (ch64ws < 0) {
    const char* e = getenv("RX_CH64WS");
    ch64ws = e ? atoi(e) : 1;
  }
  if (g.Co % 64 == 0 && TZ == 4 && TY == 4 && TX == 16 && dt != RX_F32 && ch64ws &&
      (ch64ws == 2 || (long)g.NT * (g.Co / 64) >= 256)) {
    if (in->cs) RX_FAIL(RX_EUNSUPPORTED, "conv_halo64ws: planar-concat input");
    rx_note_kernel("conv_halo64ws_kernel");
    static int st64 = -1;    // RX_CH64_STATS=0: statistics of the 64-channel-block layers by the separate pass
    if (st64 < 0) {
      const char* e = getenv("RX_CH64_STATS");
      st64 = e ? atoi(e) : 1;
    }
    const bool room64 = stat_part && stat_chunks && (size_t)g.N * 1024 * 2 * g.Co * sizeof(float) <= stat_bytes;
    static int bs64 = -1;    // RX_CH64_BWD_STATS=0: no backward sums out of this kernel
    if (bs64 < 0) {
      const char* e = getenv("RX_CH64_BWD_STATS");
      bs64 = e ? atoi(e) : 1;
    }
    const bool fuse64 = st64 && room64 && !flip && !accumulate && !bs;
    const bool fuse64b = bs64 && room64 && flip && bs;
    g.stat_part = (fuse64 || fuse64b) ? stat_part : nullptr;
    if (fuse64b) {
      g.bs_y = bs->y->ptr, g.bs_ldy = bs->y->ld, g.bs_yss = rx_act_voxels(bs->y) * (long)bs->y->ld;
      g.bs_stats = bs->stats, g.bs_slope = bs->slope;
    }
    if (dt == RX_BF16)
      ch64ws_launch<bf16_t>(st, in->ptr, w, bias, out->ptr, g);
    else
      ch64ws_launch<f16_t>(st, in->ptr, w, bias, out->ptr, g);
    hipError_t e5 = hipGetLastError();
    if (e5 != hipSuccess) {
      rx_set_error("conv_halo64ws: %s", hipGetErrorString(e5));
      return RX_ELAUNCH;
    }
    if (fuse64 && g.stat_part) *stat_chunks = g.wgs_s * 4;      // (the launcher clears stat_part when it took a variant without them)
    if (fuse64b && g.bs_y && g.stat_part) *stat_chunks = g.wgs_s * 4;
    g.stat_part = nullptr, g.bs_y = nullptr;
    return 1;
  }
  if (in->cs || out->cs) RX_FAIL(RX_EUNSUPPORTED, "conv_halo: planar-concat operand on the generic halo kernel");
  rx_note_kernel(BN == 64 ? "conv_halo_kernel<64>" : "conv_halo_kernel<32>");
  switch (dt) {
    case RX_F32: ch_dispatch<float>(BN, grid, st, in->ptr, w, bias, out->ptr, g1); break;
    case RX_BF16: ch_dispatch<bf16_t>(BN, grid, st, in->ptr, w, bias, out->ptr, g1); break;
    case RX_F16: ch_dispatch<f16_t>(BN, grid, st, in->ptr, w, bias, out->ptr, g1); break;
    default: return 0;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rx_set_error("conv_halo: %s", hipGetErrorString(e));
    return RX_ELAUNCH;
  }
  return 1;
}
