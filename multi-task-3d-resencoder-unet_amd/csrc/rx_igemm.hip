// rx_igemm.hip -- tap-table implicit GEMM on MFMA for every "forward-shaped" contraction of the
// hot path:  Conv3d forward (stride 1/2), Conv3d backward-data (stride 1: one launch; stride 2:
// one launch per output parity class), ConvTranspose3d(k=s) forward (one launch per phase) and
// backward-data.
//
//   D[co][q] = sum_{tap} sum_{ci} W[tap.w][co][ci] * In[ q*is + tap.d ][ci]      (zero outside)
//   Out[ q*os + op ][co] (+)= D[co][q] (+ bias[co])
//
// MFMA orientation: the WEIGHTS are the A operand (rows = output channels) and the gathered
// ACTIVATIONS the B operand (columns = voxels), so a lane of the 32x32 accumulator holds 4 runs of
// 4 consecutive output channels of ONE voxel -> channels-last stores of 8/16 bytes.
// fp32 mode uses v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), bf16/f16 modes 32x32x16.
//
// Tiling: 256 threads = 4 waves; block tile BM voxels x BN channels; K tile = 64 bytes of input
// channels of one tap; LDS tiles are [row][64 B] with the 16-byte chunk index XOR-swizzled by
// (row>>2)&3 (conflict-free ds_read_b128 for the 32x32 operand fetch); register-staged double
// buffering (global loads of tile k+1 in flight under the MFMAs of tile k).
// Optional split-K (deep 4^3/8^3 layers: few voxels, 14 MB of weights): fp32 slabs + a
// deterministic reduce kernel.
#include <stdlib.h>

#include "rx_common.h"


// One launch covers up to 8 PHASES that share input / output / weights but differ in iteration grid, output
// phase offset and tap list: the 8 output parity classes of a stride-2 backward-data, or the 8 kernel positions
// of a k=s=2 transposed convolution (previously 8 launches of ~10 us each).
struct IgemmPhase {
  int Qz, Qy, Qx, Vq;
  int opz, opy, opx;
  int tap0, ntaps;
  int mtile0, mtiles;
  long slab_off;  // element offset of this phase's split-K slabs
};

struct IgemmGeom {
  int Zi, Yi, Xi, Ci, ldi;
  long in_ss;
  int isz, isy, isx;
  int Zo, Yo, Xo, Co, ldo;
  long out_ss;
  int osz, osy, osx;
  int accumulate, ksplit, nph, total_mtiles;
  IgemmPhase ph[8];
  RxTap taps[RX_MAX_TAPS];
};

__device__ inline int swz(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                    T* __restrict__ out, float* __restrict__ slab, const IgemmGeom g) {
  constexpr int P = Elem<T>::PER16;  // elements per 16 B
  constexpr int KB = 4 * P;          // input channels per K tile (64 B)
  constexpr int NB = BN / 32;        // channel blocks per wave
  constexpr int MV = BM / 128;       // voxel blocks per wave
  constexpr int XP = BM / 64;        // activation pieces per thread
  constexpr int WP = (BN + 63) / 64; // weight pieces per thread
  __shared__ u32x4 sX[2][BM * 4];
  __shared__ u32x4 sW[2][BN * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt_all = blockIdx.x % g.total_mtiles, split = blockIdx.x / g.total_mtiles;
  int phase = 0;
  for (int i = 1; i < g.nph; ++i)
    if (mt_all >= g.ph[i].mtile0) phase = i;
  const IgemmPhase P_ = g.ph[phase];
  const int mtile = mt_all - P_.mtile0;
  const int m0 = mtile * BM, n0 = blockIdx.y * BN, n = blockIdx.z;
  const T* in_n = in + (long)n * g.in_ss;

  // ---- per-thread row geometry for the activation gather (fixed for the whole K loop)
  const int chunk = tid & 3;
  int rz[XP], ry[XP], rx[XP];
  bool rok[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    int q = m0 + (tid >> 2) + 64 * p;
    rok[p] = q < P_.Vq;
    int qx = q % P_.Qx, t = q / P_.Qx;
    int qy = t % P_.Qy, qz = t / P_.Qy;
    rz[p] = qz * g.isz;
    ry[p] = qy * g.isy;
    rx[p] = qx * g.isx;
  }
  const int nkc = g.Ci / KB;
  const int nkt_all = P_.ntaps * nkc;
  const int kt_begin = (int)((long)nkt_all * split / g.ksplit);
  const int kt_end = (int)((long)nkt_all * (split + 1) / g.ksplit);

  u32x4 xr[XP], wr[WP];
  auto load_tile = [&](int kt) {
    const int tap = kt / nkc, cc = kt - tap * nkc;
    const RxTap tp = g.taps[P_.tap0 + tap];
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      int z = rz[p] + tp.dz, y = ry[p] + tp.dy, x = rx[p] + tp.dx;
      bool ok = rok[p] && (unsigned)z < (unsigned)g.Zi && (unsigned)y < (unsigned)g.Yi && (unsigned)x < (unsigned)g.Xi;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(in_n + ((long)(z * g.Yi + y) * g.Xi + x) * g.ldi + cc * KB + chunk * P);
      xr[p] = v;
    }
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      int row = (tid >> 2) + 64 * p;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (row < BN) v = *reinterpret_cast<const u32x4*>(w + ((long)tp.w * g.Co + n0 + row) * g.Ci + cc * KB + chunk * P);
      wr[p] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < XP; ++p) sX[buf][swz((tid >> 2) + 64 * p, chunk)] = xr[p];
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      int row = (tid >> 2) + 64 * p;
      if (row < BN) sW[buf][swz(row, chunk)] = wr[p];
    }
  };

  f32x16 acc[NB][MV];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < MV; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[NB], bf[MV];
#pragma unroll
      for (int a = 0; a < NB; ++a) af[a] = sW[buf][swz(a * 32 + fr, ks * 2 + fh)];
#pragma unroll
      for (int b = 0; b < MV; ++b) bf[b] = sX[buf][swz(wave * (BM / 4) + b * 32 + fr, ks * 2 + fh)];
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < MV; ++b) Mma<T>::run(acc[a][b], af[a], bf[b]);
    }
    if (kt + 1 < kt_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds voxel (lane&31) of each voxel block, channel runs 8*g4 + 4*fh + (0..3)
#pragma unroll
  for (int b = 0; b < MV; ++b) {
    const int q = m0 + wave * (BM / 4) + b * 32 + fr;
    if (q >= P_.Vq) continue;
    if (g.ksplit > 1) {
      float* sp = slab + P_.slab_off + (((long)split * gridDim.z + n) * P_.Vq + q) * g.Co + n0;
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
          *reinterpret_cast<f32x4*>(sp + a * 32 + 8 * g4 + 4 * fh) = v;
        }
      continue;
    }
    int qx = q % P_.Qx, t = q / P_.Qx;
    int qy = t % P_.Qy, qz = t / P_.Qy;
    long ov = ((long)(qz * g.osz + P_.opz) * g.Yo + (qy * g.osy + P_.opy)) * g.Xo + (qx * g.osx + P_.opx);
    T* op = out + (long)n * g.out_ss + ov * g.ldo + n0;
    auto epi = [&](auto HB, auto RA) {       // bias / accumulate as compile-time flags behind one uniform branch (RX_EPI_DISPATCH)
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

// split-K reduce: out[q*os+op][c] (+)= sum_s slab[s][n][q][c] (+ bias); blockIdx.y = phase
template <typename T>
__global__ __launch_bounds__(256) void igemm_splitk_reduce(const float* __restrict__ slab, const float* __restrict__ bias, T* __restrict__ out,
                                                           const IgemmGeom g, int N) {
  const IgemmPhase P_ = g.ph[blockIdx.y];
  const int CV = g.Co / 4;
  const long total = (long)N * P_.Vq * CV;
  const float* sl = slab + P_.slab_off;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cv = (int)(i % CV);
    long nq = i / CV;
    int q = (int)(nq % P_.Vq), n = (int)(nq / P_.Vq);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < g.ksplit; ++k) s += *reinterpret_cast<const f32x4*>(sl + (((long)k * N + n) * P_.Vq + q) * g.Co + cv * 4);
    int qx = q % P_.Qx, t = q / P_.Qx;
    int qy = t % P_.Qy, qz = t / P_.Qy;
    long ov = ((long)(qz * g.osz + P_.opz) * g.Yo + (qy * g.osy + P_.opy)) * g.Xo + (qx * g.osx + P_.opx);
    T* op = out + (long)n * g.out_ss + ov * g.ldo + cv * 4;
    T vals[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = s[j];
      if (bias) f += bias[cv * 4 + j];
      if (g.accumulate) f += Elem<T>::to_f(op[j]);
      vals[j] = Elem<T>::from_f(f);
    }
    if (sizeof(T) == 2)
      *reinterpret_cast<u32x2*>(op) = *reinterpret_cast<u32x2*>(vals);
    else
      *reinterpret_cast<u32x4*>(op) = *reinterpret_cast<u32x4*>(vals);
  }
}

// ---------------------------------------------------------------------------------------------
// Low-resolution layers (8^3 / 4^3: N*V <= 2048 voxels, 512 channels, 14 MB of weights per layer).  The kernel above
// is latency bound there: a 64-byte K tile is only 4 MFMAs per wave between two barriers (~1.5 us per K step), half of a
// 128-voxel tile is padding at 4^3 (one sample = 64 voxels), and every sample re-reads the weights.  This variant
//   * folds the batch into M (row Q = n*V + v), so a weight slice is read once per M tile instead of once per sample;
//   * takes 256-byte K tiles (128 channels of one tap): 16 MFMAs per wave per barrier pair, 4x fewer steps, 4x the
//     bytes in flight per step; rows are 256 B with the 16-byte chunk index XOR-ed with (row & 15) (conflict free for
//     the ds_read_b128 lane groups: both operands' 16-lane groups cover 16 distinct row residues, see lane_voxel);
//   * always splits K (taps x channel groups) so that >= 512 workgroups exist; fp32 slabs + the reduce below.
// ---------------------------------------------------------------------------------------------
__device__ inline int lane_voxel_ig(int l) { return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l; }

template <typename T, int ABL = 0>      // ABL: timing ablations (-DRX_ABLATION=1 builds only): 1 no global loads, 2 no MFMAs, 4 no LDS traffic
__global__ __launch_bounds__(256, 2) void igemm_fat_kernel(const T* __restrict__ in, const T* __restrict__ w, float* __restrict__ slab,
                                                           const IgemmGeom g, int NV /* N * Vq */, int nsteps_total) {
  constexpr int P = Elem<T>::PER16;
  constexpr int KC = 16;             // 16-byte chunks per K row (256 B)
  constexpr int KE = KC * P;         // input channels per K step
  constexpr int BM = 128, BN = 64;
  constexpr int XP = BM * KC / 256;  // 8 pieces per thread
  constexpr int WP = BN * KC / 256;  // 4
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  u32x4* sX = reinterpret_cast<u32x4*>(smem_f);          // [BM][KC]
  u32x4* sW = sX + BM * KC;                              // [BN][KC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mtiles = (NV + BM - 1) / BM;
  // the M tiles of one (channel block, K split) read the SAME weight slice: keep them on one XCD (one L2) so that the
  // slice comes from HBM once, not once per M tile
  const int lid = rx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int mtile = lid % mtiles, rest = lid / mtiles;
  const int split = rest % g.ksplit, n0 = (rest / g.ksplit) * BN;
  const IgemmPhase P_ = g.ph[0];
  const int chunk = tid & 15, rbase = tid >> 4;          // 16 rows per pass

  // ---- per-thread gather geometry: rows rbase + 16p of the M tile
  int gz[XP], gy[XP], gx[XP];
  long gbase[XP];
  bool rok[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    const int Q = mtile * BM + rbase + 16 * p;
    rok[p] = Q < NV;
    const int n = Q / P_.Vq, q = Q - n * P_.Vq;
    const int qx = q % P_.Qx, t = q / P_.Qx;
    const int qy = t % P_.Qy, qz = t / P_.Qy;
    gz[p] = qz * g.isz, gy[p] = qy * g.isy, gx[p] = qx * g.isx;
    gbase[p] = (long)n * g.in_ss + chunk * P;
  }
  const int nkc = g.Ci / KE;
  const int st_begin = (int)((long)nsteps_total * split / g.ksplit);
  const int st_end = (int)((long)nsteps_total * (split + 1) / g.ksplit);

  // SQ counters of this kernel: 39 % of the wave cycles issuing (VALU 21 %, two waves per SIMD), MFMA busy 14 % -- it was bound by
  // the instruction stream of the gather, ~20 VALU + a branch per activation row and K step against 16 MFMAs per wave.  The row
  // offsets only change with the TAP, not with the channel chunk: they are recomputed when the tap changes (every Ci/128 steps) and
  // kept as one 32-bit element offset per row (-1 = zero padding).
  u32x4 xr[XP], wr[WP];
  int xo[XP], wo = 0, cur_tap = -1;
  auto load_step = [&](int st) {
    const int tap = st / nkc, cc = st - tap * nkc;
    if (tap != cur_tap) {                     // uniform
      cur_tap = tap;
      const RxTap tp = g.taps[P_.tap0 + tap];
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        const int z = gz[p] + tp.dz, y = gy[p] + tp.dy, x = gx[p] + tp.dx;
        const bool ok = rok[p] & ((unsigned)z < (unsigned)g.Zi) & ((unsigned)y < (unsigned)g.Yi) & ((unsigned)x < (unsigned)g.Xi);
        xo[p] = ok ? (int)(gbase[p] + ((long)(z * g.Yi + y) * g.Xi + x) * g.ldi) : -1;
      }
      wo = (int)(((long)tp.w * g.Co + n0 + rbase) * g.Ci + chunk * P);
    }
    const int ko = cc * KE;
    if (ABL & 1) {
#pragma unroll
      for (int p = 0; p < XP; ++p) xr[p] = u32x4{(unsigned)(xo[p] + ko), 1u, 2u, 3u};
#pragma unroll
      for (int p = 0; p < WP; ++p) wr[p] = u32x4{(unsigned)(wo + ko), 1u, 2u, 3u};
      return;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int o = xo[p] < 0 ? 0 : xo[p] + ko;           // always a valid address: the load is unconditional, the select zeroes padding
      const u32x4 v = *reinterpret_cast<const u32x4*>(in + o);
      xr[p] = xo[p] < 0 ? u32x4{0u, 0u, 0u, 0u} : v;
    }
#pragma unroll
    for (int p = 0; p < WP; ++p) wr[p] = *reinterpret_cast<const u32x4*>(w + wo + (long)(16 * p) * g.Ci + ko);
  };
  auto store_step = [&]() {
    if (ABL & 4) {          // keep the values alive without touching LDS
      if (xr[0][0] == 0x7fffffffu && wr[0][0] == 0x7ffffffeu) sX[0] = xr[1];
      return;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int row = rbase + 16 * p;
      sX[row * KC + (chunk ^ (row & 15))] = xr[p];
    }
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int row = rbase + 16 * p;
      sW[row * KC + (chunk ^ (row & 15))] = wr[p];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int vrow = wave * 32 + lane_voxel_ig(fr);         // this lane's voxel row of the M tile
  const u32x4* xrow = sX + vrow * KC;
  const u32x4* wrow0 = sW + fr * KC;
  const u32x4* wrow1 = sW + (32 + fr) * KC;
  const int xm = vrow & 15, wm = fr & 15;                  // (32 + fr) & 15 == fr & 15

  if (st_begin < st_end) load_step(st_begin);
  for (int st = st_begin; st < st_end; ++st) {
    __syncthreads();
    store_step();
    __syncthreads();
    if (st + 1 < st_end) load_step(st + 1);
#pragma unroll
    for (int ks = 0; ks < KC / 2; ++ks) {
      const int c = ks * 2 + fh;
      u32x4 b, a0, a1;
      if (ABL & 4) {
        b = u32x4{(unsigned)c, (unsigned)st, 2u, 3u}, a0 = b, a1 = b;
      } else {
        b = xrow[c ^ xm];
        a0 = wrow0[c ^ wm], a1 = wrow1[c ^ wm];
      }
      if (ABL & 2) {
        acc[0][ks] += __builtin_bit_cast(float, a0[0] ^ b[1]);
        acc[1][ks] += __builtin_bit_cast(float, a1[2] ^ b[3]);
      } else {
        Mma<T>::run(acc[0], a0, b);
        Mma<T>::run(acc[1], a1, b);
      }
    }
  }

  // ---- fp32 partial sums: slab[split][Q][Co]
  const int Q = mtile * BM + vrow;
  if (Q < NV) {
    float* sp = slab + ((long)split * NV + Q) * g.Co + n0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v = {acc[a][4 * g4], acc[a][4 * g4 + 1], acc[a][4 * g4 + 2], acc[a][4 * g4 + 3]};
        *reinterpret_cast<f32x4*>(sp + a * 32 + 8 * g4 + 4 * fh) = v;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

template <typename T>
static void igemm_dispatch_tile(int BM, int BN, dim3 grid, hipStream_t st, const void* in, const void* w, const float* bias, void* out,
                                void* ws, const IgemmGeom& g, int N, int ks) {
  rx_note_kernel(BM == 256 ? (BN == 64 ? "igemm_kernel<256,64>" : "igemm_kernel<256,32>")
                            : (BN == 64 ? "igemm_kernel<128,64>" : "igemm_kernel<128,32>"));
  if (BM == 256 && BN == 64)
    hipLaunchKernelGGL((igemm_kernel<T, 256, 64>), grid, dim3(256), 0, st, (const T*)in, (const T*)w, bias, (T*)out, (float*)ws, g);
  else if (BM == 256 && BN == 32)
    hipLaunchKernelGGL((igemm_kernel<T, 256, 32>), grid, dim3(256), 0, st, (const T*)in, (const T*)w, bias, (T*)out, (float*)ws, g);
  else if (BM == 128 && BN == 64)
    hipLaunchKernelGGL((igemm_kernel<T, 128, 64>), grid, dim3(256), 0, st, (const T*)in, (const T*)w, bias, (T*)out, (float*)ws, g);
  else
    hipLaunchKernelGGL((igemm_kernel<T, 128, 32>), grid, dim3(256), 0, st, (const T*)in, (const T*)w, bias, (T*)out, (float*)ws, g);
  if (ks > 1) {
    int vmax = 0;
    for (int i = 0; i < g.nph; ++i) vmax = g.ph[i].Vq > vmax ? g.ph[i].Vq : vmax;
    long total = (long)N * vmax * (g.Co / 4);
    int G = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL((igemm_splitk_reduce<T>), dim3(G, g.nph), dim3(256), 0, st, (const float*)ws, bias, (T*)out, g, N);
  }
}

static int igemm_launch(rx_dtype dt, const void* in, const void* w, const float* bias, void* out, IgemmGeom& g, int N, void* ws,
                        size_t ws_bytes, hipStream_t st) {
  const int per16 = dt == RX_F32 ? 4 : 8;
  const int KB = 4 * per16;
  if (g.Ci % KB) RX_FAIL(RX_EUNSUPPORTED, "igemm: input channels must be a multiple of %d (got %d)", KB, g.Ci);
  if (g.Co % 32) RX_FAIL(RX_EUNSUPPORTED, "igemm: output channels must be a multiple of 32 (got %d)", g.Co);
  if (g.ldi % per16 || g.ldo % 4 || ((uintptr_t)in & 15) || ((uintptr_t)out & 7) || ((uintptr_t)w & 15))
    RX_FAIL(RX_EUNSUPPORTED, "igemm: misaligned operand (ldi=%d ldo=%d)", g.ldi, g.ldo);
  int vmax = 0, nkt_max = 0;
  for (int i = 0; i < g.nph; ++i) {
    vmax = g.ph[i].Vq > vmax ? g.ph[i].Vq : vmax;
    nkt_max = g.ph[i].ntaps > nkt_max ? g.ph[i].ntaps : nkt_max;
  }
  nkt_max *= g.Ci / KB;
  if (vmax <= 0 || g.nph <= 0) return RX_OK;
  // low-resolution layers: fat K steps, batch folded into M (see igemm_fat_kernel)
  {
    static int fat_on = -1;
    if (fat_on < 0) {
      const char* e = getenv("RX_NO_FAT");
      fat_on = e ? 0 : 1;
    }
    const int KE = 16 * per16;
    const long NV = (long)N * g.ph[0].Vq;
    if (fat_on && dt != RX_F32 && g.nph == 1 && g.Ci % KE == 0 && g.Co % 64 == 0 && NV <= 2048 && ws &&
        g.ph[0].ntaps * (g.Ci / KE) >= 8 && g.ph[0].opz == 0 && g.ph[0].opy == 0 && g.ph[0].opx == 0 && g.osz == 1 && g.osy == 1 &&
        g.osx == 1) {
      const int mtiles = (int)((NV + 127) / 128);
      const int nsteps = g.ph[0].ntaps * (g.Ci / KE);
      const long base_wgs = (long)mtiles * (g.Co / 64);
      static int fat_wgs = -1;
      if (fat_wgs < 0) {
        const char* e = getenv("RX_FAT_WGS");
        fat_wgs = e ? atoi(e) : 512;
      }
      int ks = (int)((fat_wgs + base_wgs - 1) / base_wgs);
      if (ks > nsteps / 4) ks = nsteps / 4;   // >= 4 K steps per workgroup; also bounds the serial sum of the reduce
      if (ks < 1) ks = 1;
      while (ks > 1 && (size_t)ks * NV * g.Co * sizeof(float) > ws_bytes) --ks;
      if ((size_t)ks * NV * g.Co * sizeof(float) <= ws_bytes) {
        g.ksplit = ks;
        g.total_mtiles = mtiles;
        g.ph[0].mtile0 = 0, g.ph[0].mtiles = mtiles, g.ph[0].slab_off = 0;
        const size_t lds = (size_t)(128 + 64) * 256;
        rx_note_kernel("igemm_fat_kernel");
        dim3 grid(mtiles * ks, g.Co / 64);
        if (dt == RX_BF16) {
          static bool attr = false;
          if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fat_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
          }
#if RX_ABLATION
          {
            const char* ea = getenv("RX_FAT_ABL");
            const int abl = ea ? atoi(ea) : 0;
#define RX_FAT_L(A)                                                                                                                     \
  case A:                                                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fat_kernel<bf16_t, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((igemm_fat_kernel<bf16_t, A>), grid, dim3(256), lds, st, (const bf16_t*)in, (const bf16_t*)w, (float*)ws, g, (int)NV, nsteps); \
    break;
            switch (abl) {
              RX_FAT_L(1) RX_FAT_L(2) RX_FAT_L(3) RX_FAT_L(4) RX_FAT_L(5) RX_FAT_L(6) RX_FAT_L(7)
              default:
                hipLaunchKernelGGL((igemm_fat_kernel<bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)in, (const bf16_t*)w, (float*)ws, g, (int)NV, nsteps);
            }
#undef RX_FAT_L
          }
#else
          hipLaunchKernelGGL((igemm_fat_kernel<bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)in, (const bf16_t*)w, (float*)ws, g, (int)NV, nsteps);
#endif
        } else {
          static bool attr = false;
          if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fat_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
          }
          hipLaunchKernelGGL((igemm_fat_kernel<f16_t>), grid, dim3(256), lds, st, (const f16_t*)in, (const f16_t*)w, (float*)ws, g, (int)NV, nsteps);
        }
        // the reduce always runs here (even for ks == 1 the kernel only writes fp32 slabs): force its split path
        const long total = NV * (g.Co / 4);
        const int G = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        RX_DISPATCH_DTYPE(dt, T, hipLaunchKernelGGL((igemm_splitk_reduce<T>), dim3(G, 1), dim3(256), 0, st, (const float*)ws, bias, (T*)out, g, N));
        RX_CHECK_LAUNCH("igemm_fat");
        return RX_OK;
      }
    }
  }
  const int BN = (g.Co % 64 == 0) ? 64 : 32;
  const int BM = (vmax <= 128) ? 128 : 256;
  g.total_mtiles = 0;
  long vsum = 0;
  for (int i = 0; i < g.nph; ++i) {
    g.ph[i].mtile0 = g.total_mtiles;
    g.ph[i].mtiles = (g.ph[i].Vq + BM - 1) / BM;
    g.total_mtiles += g.ph[i].mtiles;
    vsum += g.ph[i].Vq;
  }
  // split-K when the natural grid cannot fill the chip and K is deep
  const long wgs = (long)g.total_mtiles * (g.Co / BN) * N;
  int ks = 1;
  if (wgs < 192 && nkt_max >= 16 && ws) {
    ks = (int)((512 + wgs - 1) / wgs);
    if (ks > nkt_max / 4) ks = nkt_max / 4;
    if (ks > 32) ks = 32;
    size_t need = (size_t)ks * N * vsum * g.Co * sizeof(float);
    while (ks > 1 && need > ws_bytes) {
      --ks;
      need = (size_t)ks * N * vsum * g.Co * sizeof(float);
    }
    if (ks < 1) ks = 1;
  }
  long off = 0;
  for (int i = 0; i < g.nph; ++i) {
    g.ph[i].slab_off = off;
    off += (long)ks * N * g.ph[i].Vq * g.Co;
  }
  g.ksplit = ks;
  dim3 grid(g.total_mtiles * ks, g.Co / BN, N);
  RX_DISPATCH_DTYPE(dt, T, igemm_dispatch_tile<T>(BM, BN, grid, st, in, w, bias, out, ws, g, N, ks));
  RX_CHECK_LAUNCH("igemm");
  return RX_OK;
}

// rx_conv_halo.hip
int rx_conv_halo_try(rx_dtype dt, const rx_act* in, const void* w, const float* bias, const rx_act* out, int flip, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st, float* stat_part, size_t stat_bytes, int* stat_chunks,
                     const RxBwdStat* bs);
void rx_inbwd_fused_finalize_launch(const float* partial, int N, int nchunks, int C, double V, const float* stats, float* m12, hipStream_t st);
// rx_pointwise.hip
int rx_pointwise_try(rx_dtype dt, const rx_act* in, const void* w, const float* bias, const rx_act* out, const int32_t stride[3],
                     int accumulate, hipStream_t st);
// rx_dgrad_s2.hip
int rx_dgrad_s2_halo_try(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx, int accumulate, hipStream_t st);
// rx_elementwise.hip
void rx_stats_finalize_launch(const float* partial, int N, int nchunks, int C, double V, float eps, float* stats, hipStream_t st);
static bool is_333_s1(const int32_t k[3], const int32_t s[3]) {
  return k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1;
}

static int check13(const int32_t k[3], const int32_t s[3], const char* who) {
  for (int i = 0; i < 3; ++i) {
    if (k[i] < 1 || k[i] > RX_MAX_KERNEL) RX_FAIL(RX_EUNSUPPORTED, "%s: kernel sizes must be 1..%d per axis (got %d)", who, RX_MAX_KERNEL, k[i]);
    if (s[i] < 1 || s[i] > RX_MAX_STRIDE) RX_FAIL(RX_EUNSUPPORTED, "%s: strides must be 1..%d per axis (got %d)", who, RX_MAX_STRIDE, s[i]);
  }
  return RX_OK;
}
static int conv_out_dim(int in, int k, int s) { return (in + 2 * ((k - 1) / 2) - k) / s + 1; }

static void geom_in(IgemmGeom& g, const rx_act* a) {
  g.Zi = a->z, g.Yi = a->y, g.Xi = a->x, g.Ci = a->c, g.ldi = a->ld;
  g.in_ss = rx_act_voxels(a) * (long)a->ld;
}
static void geom_out(IgemmGeom& g, const rx_act* a) {
  g.Zo = a->z, g.Yo = a->y, g.Xo = a->x, g.Co = a->c, g.ldo = a->ld;
  g.out_ss = rx_act_voxels(a) * (long)a->ld;
}

extern "C" int rx_conv3d_fwd(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                             const int32_t kernel[3], const int32_t stride[3], void* ws, size_t wsb, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), y_ = RxActV(y), kernel_ = RxI3V(kernel), stride_ = RxI3V(stride)](void* s) { return rx_conv3d_fwd(dt, x_.p(), w_fwd, bias, y_.p(), kernel_.v, stride_.v, ws, wsb, s); });
  if (!rx_act_ok_planar(x) || !rx_act_ok(y) || !w_fwd) RX_FAIL(RX_EINVAL, "rx_conv3d_fwd: bad arguments");
  int rc = check13(kernel, stride, "rx_conv3d_fwd");
  if (rc) return rc;
  if (y->n != x->n || y->z != conv_out_dim(x->z, kernel[0], stride[0]) || y->y != conv_out_dim(x->y, kernel[1], stride[1]) ||
      y->x != conv_out_dim(x->x, kernel[2], stride[2]))
    RX_FAIL(RX_EINVAL, "rx_conv3d_fwd: output geometry mismatch");
  if (is_333_s1(kernel, stride)) {
    rc = rx_conv_halo_try(dt, x, w_fwd, bias, y, 0, 0, ws, wsb, (hipStream_t)stream, nullptr, 0, nullptr, nullptr);  // LDS-halo kernel
    if (rc < 0) return rc;
    if (rc == 1) return RX_OK;
  }
  if (x->cs) RX_FAIL(RX_EUNSUPPORTED, "rx_conv3d_fwd: a planar-concat input needs the 3x3x3 stride-1 halo kernel (Co = 32, X >= 16)");
  if (kernel[0] == 1 && kernel[1] == 1 && kernel[2] == 1 && stride[0] == 1 && stride[1] == 1 && stride[2] == 1) {
    const int32_t one[3] = {1, 1, 1};
    if (rx_pointwise_try(dt, x, w_fwd, bias, y, one, 0, (hipStream_t)stream) == 1) {      // streaming kernel, no spatial footprint
      RX_CHECK_LAUNCH("rx_conv3d_fwd(pointwise)");
      return RX_OK;
    }
  }
  IgemmGeom g;
  memset(&g, 0, sizeof(g));
  geom_in(g, x);
  geom_out(g, y);
  g.nph = 1;
  IgemmPhase& P = g.ph[0];
  P.Qz = y->z, P.Qy = y->y, P.Qx = y->x, P.Vq = (int)rx_act_voxels(y);
  g.isz = stride[0], g.isy = stride[1], g.isx = stride[2];
  g.osz = g.osy = g.osx = 1;
  const int pz = (kernel[0] - 1) / 2, py = (kernel[1] - 1) / 2, px = (kernel[2] - 1) / 2;
  for (int a = 0; a < kernel[0]; ++a)
    for (int b = 0; b < kernel[1]; ++b)
      for (int c = 0; c < kernel[2]; ++c) {
        RxTap& t = g.taps[P.ntaps];
        t.dz = (int8_t)(a - pz), t.dy = (int8_t)(b - py), t.dx = (int8_t)(c - px);
        t.w = (uint16_t)P.ntaps;
        ++P.ntaps;
      }
  return igemm_launch(dt, x->ptr, w_fwd, bias, y->ptr, g, x->n, ws, wsb, (hipStream_t)stream);
}

// backward-data + the two sums of the InstanceNorm backward of the layer whose output gradient this launch COMPLETES (dx =
// dL/d(out of that layer); the layer has no residual, its LeakyReLU mask is the sign of the normalised value).  *fused = 1 and
// m12[n][c] = (mean g', mean g'*xhat) when the launch ran on the persistent 32-channel kernel (continue with
// rx_instnorm_act_bwd_apply); otherwise *fused = 0, m12 untouched (continue with rx_instnorm_act_bwd).  The caller guarantees
// that nothing adds to dx afterwards.
extern "C" int rx_conv3d_bwd_data_instats(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx, const int32_t kernel[3],
                                          const int32_t stride[3], int accumulate, const rx_act* in_y, const float* in_stats, float slope,
                                          float* m12, int* fused, void* ws, size_t wsb, void* stream) {
  RX_RECORD(stream, [=, dy_ = RxActV(dy), dx_ = RxActV(dx), kernel_ = RxI3V(kernel), stride_ = RxI3V(stride), in_y_ = RxActV(in_y)](void* s) { int fused_dummy = 0; return rx_conv3d_bwd_data_instats(dt, dy_.p(), w_bwd, dx_.p(), kernel_.v, stride_.v, accumulate, in_y_.p(), in_stats, slope, m12, &fused_dummy, ws, wsb, s); });
  if (!fused || !m12 || !in_stats || !rx_act_ok(in_y) || !ws) RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_data_instats: bad arguments");
  *fused = 0;
  if (!rx_act_ok(dy) || !rx_act_ok(dx) || !w_bwd) RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_data_instats: bad arguments");
  int rc = check13(kernel, stride, "rx_conv3d_bwd_data_instats");
  if (rc) return rc;
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("RX_FUSED_BWD_STATS");
    on = e ? atoi(e) : 1;
  }
  const bool same = in_y->n == dx->n && in_y->z == dx->z && in_y->y == dx->y && in_y->x == dx->x && in_y->c == dx->c;
  if (on && same && is_333_s1(kernel, stride) && dx->z == dy->z && dx->y == dy->y && dx->x == dy->x && in_y->ld % 4 == 0 &&
      !((uintptr_t)in_y->ptr & 7)) {
    RxBwdStat bs{in_y, in_stats, slope};
    int chunks = 0;
    rc = rx_conv_halo_try(dt, dy, w_bwd, nullptr, dx, 1, accumulate, nullptr, 0, (hipStream_t)stream, (float*)ws, wsb, &chunks, &bs);
    if (rc < 0) return rc;
    if (rc == 1) {
      if (chunks > 0) {
        rx_inbwd_fused_finalize_launch((const float*)ws, dx->n, chunks, dx->c, (double)rx_act_voxels(dx), in_stats, m12, (hipStream_t)stream);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) RX_FAIL(RX_ELAUNCH, "rx_conv3d_bwd_data_instats(finalize): %s", hipGetErrorString(e));
        *fused = 1;
      }
      return RX_OK;
    }
  }
  return rx_conv3d_bwd_data(dt, dy, w_bwd, dx, kernel, stride, accumulate, ws, wsb, stream);
}

// conv + InstanceNorm statistics of its output in one call.  When the layer runs on a persistent halo kernel the sums of y and
// y^2 come out of the conv epilogue (no extra pass over y); otherwise this is rx_conv3d_fwd followed by rx_instnorm_stats.
extern "C" int rx_conv3d_fwd_stats(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                                   const int32_t kernel[3], const int32_t stride[3], float eps, float* stats, void* ws, size_t wsb,
                                   void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), y_ = RxActV(y), kernel_ = RxI3V(kernel), stride_ = RxI3V(stride)](void* s) { return rx_conv3d_fwd_stats(dt, x_.p(), w_fwd, bias, y_.p(), kernel_.v, stride_.v, eps, stats, ws, wsb, s); });
  if (!rx_act_ok_planar(x) || !rx_act_ok(y) || !w_fwd || !stats || !ws) RX_FAIL(RX_EINVAL, "rx_conv3d_fwd_stats: bad arguments");
  int rc = check13(kernel, stride, "rx_conv3d_fwd_stats");
  if (rc) return rc;
  if (is_333_s1(kernel, stride) && y->n == x->n && y->z == x->z && y->y == x->y && y->x == x->x) {
    static int fuse = -1;
    if (fuse < 0) {
      const char* e = getenv("RX_FUSED_STATS");
      fuse = e ? atoi(e) : 1;
    }
    int chunks = 0;
    rc = rx_conv_halo_try(dt, x, w_fwd, bias, y, 0, 0, nullptr, 0, (hipStream_t)stream, fuse ? (float*)ws : nullptr, wsb, &chunks, nullptr);
    if (rc < 0) return rc;
    if (rc == 1 && chunks > 0) {
      rx_stats_finalize_launch((const float*)ws, y->n, chunks, y->c, (double)rx_act_voxels(y), eps, stats, (hipStream_t)stream);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) RX_FAIL(RX_ELAUNCH, "rx_conv3d_fwd_stats(finalize): %s", hipGetErrorString(e));
      return RX_OK;
    }
    if (rc == 1) return rx_instnorm_stats(dt, y, eps, stats, ws, wsb, stream);
  }
  rc = rx_conv3d_fwd(dt, x, w_fwd, bias, y, kernel, stride, ws, wsb, stream);
  if (rc) return rc;
  return rx_instnorm_stats(dt, y, eps, stats, ws, wsb, stream);
}

extern "C" int rx_conv3d_bwd_data(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx, const int32_t kernel[3],
                                  const int32_t stride[3], int accumulate, void* ws, size_t wsb, void* stream) {
  RX_RECORD(stream, [=, dy_ = RxActV(dy), dx_ = RxActV(dx), kernel_ = RxI3V(kernel), stride_ = RxI3V(stride)](void* s) { return rx_conv3d_bwd_data(dt, dy_.p(), w_bwd, dx_.p(), kernel_.v, stride_.v, accumulate, ws, wsb, s); });
  if (!rx_act_ok(dy) || !rx_act_ok_planar(dx) || !w_bwd) RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_data: bad arguments");
  int rc = check13(kernel, stride, "rx_conv3d_bwd_data");
  if (rc) return rc;
  if (dy->n != dx->n || dy->z != conv_out_dim(dx->z, kernel[0], stride[0]) || dy->y != conv_out_dim(dx->y, kernel[1], stride[1]) ||
      dy->x != conv_out_dim(dx->x, kernel[2], stride[2]))
    RX_FAIL(RX_EINVAL, "rx_conv3d_bwd_data: geometry mismatch");
  if (is_333_s1(kernel, stride)) {
    rc = rx_conv_halo_try(dt, dy, w_bwd, nullptr, dx, 1, accumulate, ws, wsb, (hipStream_t)stream, nullptr, 0, nullptr, nullptr);
    if (rc < 0) return rc;
    if (rc == 1) return RX_OK;
  }
  if (dx->cs) RX_FAIL(RX_EUNSUPPORTED, "rx_conv3d_bwd_data: a planar-concat dx needs the wave-specialised 64-channel halo kernel");
  if (kernel[0] == 1 && kernel[1] == 1 && kernel[2] == 1 && stride[0] == 1 && stride[1] == 1 && stride[2] == 1) {
    const int32_t one[3] = {1, 1, 1};
    if (rx_pointwise_try(dt, dy, w_bwd, nullptr, dx, one, accumulate, (hipStream_t)stream) == 1) {
      RX_CHECK_LAUNCH("rx_conv3d_bwd_data(pointwise)");
      return RX_OK;
    }
  }
  if (kernel[0] == 3 && kernel[1] == 3 && kernel[2] == 3 && stride[0] == 2 && stride[1] == 2 && stride[2] == 2) {
    rc = rx_dgrad_s2_halo_try(dt, dy, w_bwd, dx, accumulate, (hipStream_t)stream);   // LDS-halo kernel, all 8 parity classes
    if (rc < 0) return rc;
    if (rc == 1) return RX_OK;
  }
  const int k[3] = {kernel[0], kernel[1], kernel[2]}, s[3] = {stride[0], stride[1], stride[2]};
  const int p[3] = {(k[0] - 1) / 2, (k[1] - 1) / 2, (k[2] - 1) / 2};
  const int din[3] = {dx->z, dx->y, dx->x};
  // dx[s*q + r] = sum_{t : (r+p-t) % s == 0} dy[q + (r+p-t)/s] * W[t]^T      per axis; one PHASE per parity class r.
  // Up to 8 phases and RX_MAX_TAPS taps per launch: stride 2 is one launch (8 classes, 27 taps), strides 3 / 4 or 5..7-wide
  // kernels take a few (the classes write disjoint voxels, so the launches are independent).
  IgemmGeom g;
  auto reset = [&]() {
    memset(&g, 0, sizeof(g));
    geom_in(g, dy);
    geom_out(g, dx);
    g.isz = g.isy = g.isx = 1;
    g.osz = s[0], g.osy = s[1], g.osx = s[2];
    g.accumulate = accumulate;
  };
  reset();
  int ntap_total = 0;
  for (int r0 = 0; r0 < s[0]; ++r0)
    for (int r1 = 0; r1 < s[1]; ++r1)
      for (int r2 = 0; r2 < s[2]; ++r2) {
        const int r[3] = {r0, r1, r2};
        int Q[3];
        for (int a = 0; a < 3; ++a) Q[a] = (din[a] - r[a] + s[a] - 1) / s[a];
        if (Q[0] * Q[1] * Q[2] <= 0) continue;
        int cnt = 0;
        for (int a = 0; a < k[0]; ++a)
          if ((r0 + p[0] - a) % s[0] == 0)
            for (int b = 0; b < k[1]; ++b)
              if ((r1 + p[1] - b) % s[1] == 0)
                for (int c = 0; c < k[2]; ++c)
                  if ((r2 + p[2] - c) % s[2] == 0) ++cnt;
        if (g.nph == 8 || ntap_total + cnt > RX_MAX_TAPS) {
          rc = igemm_launch(dt, dy->ptr, w_bwd, nullptr, dx->ptr, g, dx->n, ws, wsb, (hipStream_t)stream);
          if (rc) return rc;
          reset();
          ntap_total = 0;
        }
        IgemmPhase& P = g.ph[g.nph++];
        P.Qz = Q[0], P.Qy = Q[1], P.Qx = Q[2], P.Vq = Q[0] * Q[1] * Q[2];
        P.opz = r0, P.opy = r1, P.opx = r2;
        P.tap0 = ntap_total;
        // a class without any tap (kernel narrower than the stride) still has to WRITE its voxels: zeros (or keep dx when accumulating)
        for (int a = 0; a < k[0]; ++a) {
          if ((r0 + p[0] - a) % s[0]) continue;
          for (int b = 0; b < k[1]; ++b) {
            if ((r1 + p[1] - b) % s[1]) continue;
            for (int c = 0; c < k[2]; ++c) {
              if ((r2 + p[2] - c) % s[2]) continue;
              RxTap& t = g.taps[ntap_total++];
              t.dz = (int8_t)((r0 + p[0] - a) / s[0]);
              t.dy = (int8_t)((r1 + p[1] - b) / s[1]);
              t.dx = (int8_t)((r2 + p[2] - c) / s[2]);
              t.w = (uint16_t)((a * k[1] + b) * k[2] + c);
              ++P.ntaps;
            }
          }
        }
      }
  return igemm_launch(dt, dy->ptr, w_bwd, nullptr, dx->ptr, g, dx->n, ws, wsb, (hipStream_t)stream);
}

static int checkT(const int32_t s[3], const rx_act* small, const rx_act* big, const char* who) {
  for (int i = 0; i < 3; ++i)
    if (s[i] < 1 || s[i] > RX_MAX_STRIDE) RX_FAIL(RX_EUNSUPPORTED, "%s: strides must be 1..%d per axis", who, RX_MAX_STRIDE);
  if (small->n != big->n || big->z != small->z * s[0] || big->y != small->y * s[1] || big->x != small->x * s[2])
    RX_FAIL(RX_EINVAL, "%s: geometry mismatch", who);
  return RX_OK;
}

extern "C" int rx_convT3d_fwd(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                              const int32_t stride[3], void* ws, size_t wsb, void* stream) {
  RX_RECORD(stream, [=, x_ = RxActV(x), y_ = RxActV(y), stride_ = RxI3V(stride)](void* s) { return rx_convT3d_fwd(dt, x_.p(), w_fwd, bias, y_.p(), stride_.v, ws, wsb, s); });
  if (!rx_act_ok(x) || !rx_act_ok(y) || !w_fwd) RX_FAIL(RX_EINVAL, "rx_convT3d_fwd: bad arguments");
  int rc = checkT(stride, x, y, "rx_convT3d_fwd");
  if (rc) return rc;
  if (rx_pointwise_try(dt, x, w_fwd, bias, y, stride, 0, (hipStream_t)stream) == 1) {       // x once, y once (rx_pointwise.hip)
    RX_CHECK_LAUNCH("rx_convT3d_fwd(pointwise)");
    return RX_OK;
  }
  // y[i*s + t] = sum_ci x[i] W[ci][co][t] + b : one single-tap PHASE per kernel position t, up to 8 per launch
  IgemmGeom g;
  auto reset = [&]() {
    memset(&g, 0, sizeof(g));
    geom_in(g, x);
    geom_out(g, y);
    g.isz = g.isy = g.isx = 1;
    g.osz = stride[0], g.osy = stride[1], g.osx = stride[2];
  };
  reset();
  for (int a = 0; a < stride[0]; ++a)
    for (int b = 0; b < stride[1]; ++b)
      for (int c = 0; c < stride[2]; ++c) {
        if (g.nph == 8) {
          rc = igemm_launch(dt, x->ptr, w_fwd, bias, y->ptr, g, x->n, ws, wsb, (hipStream_t)stream);
          if (rc) return rc;
          reset();
        }
        IgemmPhase& P = g.ph[g.nph];
        P.Qz = x->z, P.Qy = x->y, P.Qx = x->x, P.Vq = (int)rx_act_voxels(x);
        P.opz = a, P.opy = b, P.opx = c;
        P.tap0 = g.nph, P.ntaps = 1;
        RxTap& t = g.taps[g.nph];
        t.dz = t.dy = t.dx = 0;
        t.w = (uint16_t)((a * stride[1] + b) * stride[2] + c);
        ++g.nph;
      }
  return igemm_launch(dt, x->ptr, w_fwd, bias, y->ptr, g, x->n, ws, wsb, (hipStream_t)stream);
}

extern "C" int rx_convT3d_bwd_data(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx, const int32_t stride[3],
                                   int accumulate, void* ws, size_t wsb, void* stream) {
  RX_RECORD(stream, [=, dy_ = RxActV(dy), dx_ = RxActV(dx), stride_ = RxI3V(stride)](void* s) { return rx_convT3d_bwd_data(dt, dy_.p(), w_bwd, dx_.p(), stride_.v, accumulate, ws, wsb, s); });
  if (!rx_act_ok(dy) || !rx_act_ok(dx) || !w_bwd) RX_FAIL(RX_EINVAL, "rx_convT3d_bwd_data: bad arguments");
  int rc = checkT(stride, dx, dy, "rx_convT3d_bwd_data");
  if (rc) return rc;
  // dx[i] = sum_t dy[i*s + t] W[t]^T
  IgemmGeom g;
  memset(&g, 0, sizeof(g));
  geom_in(g, dy);
  geom_out(g, dx);
  g.nph = 1;
  IgemmPhase& P = g.ph[0];
  P.Qz = dx->z, P.Qy = dx->y, P.Qx = dx->x, P.Vq = (int)rx_act_voxels(dx);
  g.isz = stride[0], g.isy = stride[1], g.isx = stride[2];
  g.osz = g.osy = g.osx = 1;
  g.accumulate = accumulate;
  for (int a = 0; a < stride[0]; ++a)
    for (int b = 0; b < stride[1]; ++b)
      for (int c = 0; c < stride[2]; ++c) {
        RxTap& t = g.taps[P.ntaps];
        t.dz = (int8_t)a, t.dy = (int8_t)b, t.dx = (int8_t)c;
        t.w = (uint16_t)P.ntaps;
        ++P.ntaps;
      }
  return igemm_launch(dt, dy->ptr, w_bwd, nullptr, dx->ptr, g, dx->n, ws, wsb, (hipStream_t)stream);
}

extern "C" size_t rx_conv_workspace_hint(void) { return (size_t)96 << 20; }
