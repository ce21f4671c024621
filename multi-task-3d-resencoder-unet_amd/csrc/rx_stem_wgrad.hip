// rx_stem_wgrad.hip -- forward (stem_fwd_mfma_kernel, end of file) and weight gradient of the stem convolution (Cin <= 4 input channels, NCDHW fp32 image) on MFMA,
// 16-bit compute types:   dW[co][ci][t] = sum_v dY[v][co] * x[ci][v + t - 1]
//
// GEMM view: M = Cout (32-row blocks), N = 27 taps (padded to 32 columns), K = voxels.  A = dY^T comes from the
// LDS tile [256 voxels][Cout] through ds_read_b64_tr_b16 (as in rx_wgrad_halo.hip); the B fragment of lane
// (tap, k-half) is 8 consecutive x-values of the fp32 halo tile shifted by the tap, converted to the compute
// dtype in registers.  Tile 4x4x16 voxels, 4 waves take 4 k-steps each, accumulators persist over all tiles of a
// workgroup; partials [block][ci][slot=a*9+b*3+c][co] are summed by stem_wgrad_finalize (rx_elementwise.hip).
// (fp32 mode keeps the exact VALU kernel of rx_elementwise.hip.)
#include "rx_common.h"

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_s;

template <typename T>
__device__ inline u32x4 pack8(const float (&f)[8]) {
  T v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = Elem<T>::from_f(f[j]);
  return *reinterpret_cast<u32x4*>(v);
}

template <typename T, int NB>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, int Cin, int N, int Z, int Y, int X,
                                                              const T* __restrict__ dy, int ldy, long sy, int kz, int ky, int kx,
                                                              int tiles_per_block, float* __restrict__ partial) {
  constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2, HV = (TZ + 2) * HY * HX;  // 648
  constexpr int CO = NB * 32;
  __shared__ __attribute__((aligned(16))) T sG[256 * CO];
  __shared__ float sX[4][HV + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tz_n = (Z + TZ - 1) / TZ, ty_n = (Y + TY - 1) / TY, tx_n = (X + TX - 1) / TX;
  const int NT = N * tz_n * ty_n * tx_n;
  const int t_begin = blockIdx.x * tiles_per_block, t_end = min(NT, t_begin + tiles_per_block);
  const long V = (long)Z * Y * X;
  const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;

  // B-operand geometry: column = tap slot (a*9+b*3+c), valid if inside the kernel extent
  const int tap = lane & 31, h = lane >> 5;
  const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
  const bool tap_ok = tap < 27 && ta < kz && tb < ky && tc < kx;
  const int toff = tap_ok ? ((ta - pz) * HY + (tb - py)) * HX + (tc - px) : 0;
  // A-operand geometry (tr read)
  const int g16 = lane >> 4, half = g16 & 1, ah = g16 >> 1, l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3;

  f32x16 acc[4][NB];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][a][r] = 0.f;

  // the next tile's dY pieces and image halo sit in registers while this tile's MFMAs run (the two barriers of a tile
  // used to enclose the whole global-load latency)
  constexpr int GP = CO / 8;                       // 16-byte dY pieces per thread
  constexpr int XP = (HV + 255) / 256;             // halo values per thread and input channel
  u32x4 gq[GP];
  float xq[4][XP];
  auto prefetch = [&](int tile) {
    int tx = tile % tx_n, t1 = tile / tx_n;
    int ty = t1 % ty_n, t2 = t1 / ty_n;
    int tz = t2 % tz_n, n = t2 / tz_n;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
#pragma unroll
    for (int p = 0; p < GP; ++p) {
      const int i = tid + 256 * p;
      const int v = i / (CO / 8), cv = i - v * (CO / 8);
      const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), xx = x0 + (v & 15);
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (z < Z && y < Y && xx < X) val = *reinterpret_cast<const u32x4*>(dy + n * sy + ((long)(z * Y + y) * X + xx) * ldy + cv * 8);
      gq[p] = val;
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      if (ci < Cin) {
        const float* xc = x + ((long)n * Cin + ci) * V;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
          const int i = tid + 256 * p;
          const int hx = i % HX, t = i / HX, hy = t % HY, hz = t / HY;
          const int z = z0 + hz - 1, y = y0 + hy - 1, xx = x0 + hx - 1;
          float v = 0.f;
          if (i < HV && (unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)xx < (unsigned)X) v = xc[((long)z * Y + y) * X + xx];
          xq[ci][p] = v;
        }
      }
    }
  };
  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    // dY tile: 256 voxels x CO channels, panel layout [32-channel panel][voxel][32]
#pragma unroll
    for (int p = 0; p < GP; ++p) {
      const int i = tid + 256 * p;
      const int v = i / (CO / 8), c = (i - v * (CO / 8)) * 8;
      *reinterpret_cast<u32x4*>(sG + ((c >> 5) * 256 + v) * 32 + (c & 31)) = gq[p];
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
      if (ci < Cin)
#pragma unroll
        for (int p = 0; p < XP; ++p)
          if (tid + 256 * p < HV) sX[ci][tid + 256 * p] = xq[ci][p];
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
    // this wave's 4 k-steps (16 voxels each = one x-row of the tile)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = wave * 4 + kk;                 // k-step = tile row (z = s>>2, y = s&3)
      const int hrow = (((s >> 2) + 1) * HY + ((s & 3) + 1)) * HX + 1;  // halo index of x = 0 of that row
      u32x4 af[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const T* p0 = sG + (a * 256 + 16 * s + 8 * ah + q4) * 32 + 16 * half + 4 * p4;
        s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_s*)(p0));
        s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_s*)(p0 + 4 * 32));
        u32x2 lo = __builtin_bit_cast(u32x2, t0), hi = __builtin_bit_cast(u32x2, t1);
        af[a] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
      for (int ci = 0; ci < Cin; ++ci) {
        float f[8];
        const float* bp = &sX[ci][hrow + 8 * h + toff];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = tap_ok ? bp[j] : 0.f;
        const u32x4 bf = pack8<T>(f);
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          if (ci == 0) Mma<T>::run(acc[0][a], af[a], bf);
          if (ci == 1) Mma<T>::run(acc[1][a], af[a], bf);
          if (ci == 2) Mma<T>::run(acc[2][a], af[a], bf);
          if (ci == 3) Mma<T>::run(acc[3][a], af[a], bf);
        }
      }
    }
  }
  // combine the 4 waves through LDS (reuse sG as float scratch is too small for NB=2: use a dedicated buffer)
  __shared__ float red[4][32][33];
  const int col = lane & 31, fh = lane >> 5;
  for (int ci = 0; ci < Cin; ++ci)
    for (int a = 0; a < NB; ++a) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;   // row = output channel, col = tap slot
        float v = ci == 0 ? acc[0][a][r] : ci == 1 ? acc[1][a][r] : ci == 2 ? acc[2][a][r] : acc[3][a][r];
        red[wave][row][col] = v;
      }
      __syncthreads();
      for (int i = tid; i < 27 * 32; i += 256) {
        const int t = i >> 5, co = i & 31;
        float s = red[0][co][t] + red[1][co][t] + red[2][co][t] + red[3][co][t];
        partial[(((size_t)blockIdx.x * Cin + ci) * 27 + t) * CO + a * 32 + co] = s;
      }
    }
}

// returns 1 if handled (16-bit dtypes, Cout 32 or 64), 0 otherwise; nblocks_out = number of partial blocks written
int rx_stem_wgrad_mfma_try(rx_dtype dt, const float* x, int n, int cin, int z, int y, int xx, const rx_act* dy, const int32_t kernel[3],
                           float* partial, int max_blocks, int* nblocks_out, hipStream_t st) {
  if (dt == RX_F32 || (dy->c != 32 && dy->c != 64) || dy->ld % 8 || ((uintptr_t)dy->ptr & 15) || cin > 4) return 0;
  const int NT = n * ((z + 3) / 4) * ((y + 3) / 4) * ((xx + 15) / 16);
  int blocks = NT < max_blocks ? NT : max_blocks;
  if (blocks > 768) blocks = 768;       // 3 resident workgroups per CU (44 KB of LDS each): one wave of the chip, no tail
  const int per = (NT + blocks - 1) / blocks;
  blocks = (NT + per - 1) / per;
  const long sy = rx_act_voxels(dy) * (long)dy->ld;
#define RX_SW(TT, NBB)                                                                                                           \
  hipLaunchKernelGGL((stem_wgrad_mfma_kernel<TT, NBB>), dim3(blocks), dim3(256), 0, st, x, cin, n, z, y, xx, (const TT*)dy->ptr, \
                     dy->ld, sy, kernel[0], kernel[1], kernel[2], per, partial)
  if (dt == RX_BF16) {
    if (dy->c == 32) RX_SW(bf16_t, 1); else RX_SW(bf16_t, 2);
  } else {
    if (dy->c == 32) RX_SW(f16_t, 1); else RX_SW(f16_t, 2);
  }
#undef RX_SW
  *nblocks_out = blocks;
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------------
// Stem FORWARD on MFMA (16-bit compute types, Cout = 32):  out[v][co] = bias[co] + sum_k W[co][k] * X[k][v],  k = (ci, tap).
// The VALU kernel (rx_elementwise.hip: one thread per voxel, 864 FMAs and 216 LDS broadcast reads per voxel) runs at ~200 us
// for the cfg2 stem although it only has to write 268 MB; here a 32-voxel block is two MFMAs (K = 27 padded to 32 for one
// input channel): A = the weights, converted once per workgroup and kept in registers; B = 8 consecutive k of the lane's voxel
// gathered from the fp32 halo tile in LDS (tap offsets from a small LDS table) and converted in registers.  Tile 4x4x16.
// ---------------------------------------------------------------------------------------------------------------------
#define RX_STEM_NOTAP 0x40000000
// STATS: the InstanceNorm statistics of the output come out of the same pass (running per-lane sums of the values as stored,
// one partial row per wave at the end -- ch_stat_flush, rx_common.h); a block's tile range then never straddles samples
// (blocks_per_sample blocks each).
// KSM: compile-time bound on the 16-wide k-steps (2 for Cin * taps <= 32 -- one input channel, 3x3x3 -- else 7): the weight
// fragments are registers, and with 7 of them the STATS variant fell from 4 to 3 resident workgroups per CU (219 us vs 134).
template <typename T, bool STATS = false, int KSM = 7>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const float* __restrict__ x, int Cin, int N, int Z, int Y, int X,
                                                            const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                                                            int ldo, long so, int kz, int ky, int kx, int tiles_per_block,
                                                            int blocks_per_sample = 0, float* __restrict__ stat_part = nullptr) {
  constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2, HV = (TZ + 2) * HY * HX;  // 648
  constexpr int XP = (HV + 255) / 256;
  __shared__ float sX[4][HV + 8];
  __shared__ int sOff[128];                       // k -> ci * (HV + 8) + tap offset inside the halo tile, or RX_STEM_NOTAP
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TT = kz * ky * kx, K = Cin * TT, KS = (K + 15) / 16;
  const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;
  const int tz_n = (Z + TZ - 1) / TZ, ty_n = (Y + TY - 1) / TY, tx_n = (X + TX - 1) / TX;
  const int NT = N * tz_n * ty_n * tx_n;
  int t_begin = blockIdx.x * tiles_per_block, t_end = min(NT, t_begin + tiles_per_block);
  const int sn = STATS ? blockIdx.x / blocks_per_sample : 0, sl = STATS ? blockIdx.x - sn * blocks_per_sample : 0;
  if (STATS) {
    const int NTs = NT / N;
    t_begin = sn * NTs + sl * tiles_per_block, t_end = min((sn + 1) * NTs, t_begin + tiles_per_block);
  }
  float s1[1][16], s2[1][16];
  if (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[0][r] = 0.f, s2[0][r] = 0.f;
  }
  const long V = (long)Z * Y * X;
  if (tid < 128) {
    int o = RX_STEM_NOTAP;
    if (tid < K) {
      const int ci = tid / TT, t = tid - ci * TT;
      const int a = t / (ky * kx), b = (t / kx) % ky, c = t % kx;
      o = ci * (HV + 8) + ((a - pz) * HY + (b - py)) * HX + (c - px);
    }
    sOff[tid] = o;
  }
  // halo element(s) of this thread: (hz, hy, hx) packed, decoded once
  int hh[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    const int i = tid + 256 * p;
    const int hx = i % HX, t = i / HX;
    hh[p] = i < HV ? ((t / HY) << 16) | ((t % HY) << 8) | hx : -1;
  }
  // A fragments: lane (co = lane & 31, k-half = lane >> 5) holds W[co][ks*16 + 8*h .. +7]   (w is (32, Cin, TT) = [co][k])
  const int fr = lane & 31, fh = lane >> 5;
  u32x4 af[KSM];
#pragma unroll
  for (int ks = 0; ks < KSM; ++ks) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ks * 16 + fh * 8 + j;
      f[j] = (ks < KS && k < K) ? w[(long)fr * K + k] : 0.f;
    }
    af[ks] = pack8<T>(f);
  }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[8 * (r >> 2) + 4 * fh + (r & 3)] : 0.f;
  float xq[4][XP];
  auto prefetch = [&](int tile) {
    const int tx = tile % tx_n, t1 = tile / tx_n, ty = t1 % ty_n, t2 = t1 / ty_n, tz = t2 % tz_n, n = t2 / tz_n;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
      if (ci < Cin) {
        const float* xc = x + ((long)n * Cin + ci) * V;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
          const int z = z0 + (hh[p] >> 16) - 1, y = y0 + ((hh[p] >> 8) & 255) - 1, xx = x0 + (hh[p] & 255) - 1;
          float v = 0.f;
          if (hh[p] >= 0 && (unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)xx < (unsigned)X) v = xc[((long)z * Y + y) * X + xx];
          xq[ci][p] = v;
        }
      }
  };
  if (t_begin < t_end) prefetch(t_begin);
  const float* sx0 = &sX[0][0];
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                               // previous tile's operand reads are done
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
      if (ci < Cin)
#pragma unroll
        for (int p = 0; p < XP; ++p)
          if (hh[p] >= 0) sX[ci][tid + 256 * p] = xq[ci][p];
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
    const int tx = tile % tx_n, t1 = tile / tx_n, ty = t1 % ty_n, t2 = t1 / ty_n, tz = t2 % tz_n, n = t2 / tz_n;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int v = (wave * 2 + b) * 32 + fr;      // this lane's voxel (the accumulator column)
      const int vx = v & 15, vy = (v >> 4) & 3, vz = v >> 6;
      const int centre = ((vz + 1) * HY + (vy + 1)) * HX + vx + 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv[r];
#pragma unroll
      for (int ks = 0; ks < KSM; ++ks) {
        if (ks < KS) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int o = sOff[ks * 16 + fh * 8 + j];
            f[j] = o != RX_STEM_NOTAP ? sx0[centre + o] : 0.f;
          }
          Mma<T>::run(acc, af[ks], pack8<T>(f));
        }
      }
      const int z = z0 + vz, y = y0 + vy, xx = x0 + vx;
      if (z >= Z || y >= Y || xx >= X) continue;
      T* op = out + n * so + ((long)(z * Y + y) * X + xx) * ldo;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        T vals[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vals[i] = Elem<T>::from_f(acc[4 * g4 + i]);
          if (STATS) {
            const float r = Elem<T>::to_f(vals[i]);
            s1[0][4 * g4 + i] += r;
            s2[0][4 * g4 + i] += r * r;
          }
        }
        *reinterpret_cast<u32x2*>(op + 8 * g4 + 4 * fh) = *reinterpret_cast<u32x2*>(vals);
      }
    }
  }
  if (STATS) ch_stat_flush<1>(s1, s2, stat_part, sn, blocks_per_sample * 4, sl * 4 + wave, 32, 0, lane);
}

// returns 1 if handled (16-bit dtypes, Cout == 32, Cin * taps <= 112), 0 otherwise.  stat_part (optional, stat_bytes large):
// per-wave partial sums of y and y^2 are left there ([n][*stat_chunks][2][32] floats) and *stat_chunks > 0
int rx_stem_fwd_mfma_try(rx_dtype dt, const float* x, int n, int cin, int z, int y, int xx, const float* w, const float* bias,
                         const rx_act* out, const int32_t kernel[3], hipStream_t st, float* stat_part, size_t stat_bytes, int* stat_chunks) {
  if (stat_chunks) *stat_chunks = 0;
  const int K = cin * kernel[0] * kernel[1] * kernel[2];
  // (the kernel stages at most 4 input channels: with 5-8 of them a small kernel -- K = Cin * taps <= 112 -- used to be accepted
  // here and computed garbage; found by tests/test_fuzz_gpu.py)
  if (dt == RX_F32 || out->c != 32 || out->ld % 4 || ((uintptr_t)out->ptr & 7) || K > 112 || cin > 4) return 0;
  const int NT = n * ((z + 3) / 4) * ((y + 3) / 4) * ((xx + 15) / 16);
  static int maxb = -1;
  if (maxb < 0) {
    const char* e = getenv("RX_STEM_BLOCKS");
    maxb = e ? atoi(e) : 1024;   // measured: 512 -> 162 us, 768 -> 131, 1024 -> 119, 1280 -> 147, 2048 -> 127, one tile per workgroup -> 193
  }
  int blocks = NT < maxb ? NT : maxb;       // 4 resident workgroups per CU, the weights / tap table are set up once per workgroup
  int per = (NT + blocks - 1) / blocks;
  blocks = (NT + per - 1) / per;
  const long so = rx_act_voxels(out) * (long)out->ld;
  const int NTs = NT / n;
  int bps = blocks / n > 0 ? blocks / n : 1;                 // statistics: whole blocks per sample
  const int per_s = (NTs + bps - 1) / bps;
  bps = (NTs + per_s - 1) / per_s;
  const bool stats = stat_part && stat_chunks && (size_t)n * bps * 4 * 2 * 32 * sizeof(float) <= stat_bytes;
#define RX_STEM_FWD(TT, KSM)                                                                                                           \
  do {                                                                                                                                 \
    if (stats)                                                                                                                         \
      hipLaunchKernelGGL((stem_fwd_mfma_kernel<TT, true, KSM>), dim3(bps * n), dim3(256), 0, st, x, cin, n, z, y, xx, w, bias,         \
                         (TT*)out->ptr, out->ld, so, kernel[0], kernel[1], kernel[2], per_s, bps, stat_part);                          \
    else                                                                                                                               \
      hipLaunchKernelGGL((stem_fwd_mfma_kernel<TT, false, KSM>), dim3(blocks), dim3(256), 0, st, x, cin, n, z, y, xx, w, bias,         \
                         (TT*)out->ptr, out->ld, so, kernel[0], kernel[1], kernel[2], per, 0, (float*)nullptr);                        \
  } while (0)
  if (dt == RX_BF16) {
    if (K <= 32) RX_STEM_FWD(bf16_t, 2);
    else RX_STEM_FWD(bf16_t, 7);
  } else {
    if (K <= 32) RX_STEM_FWD(f16_t, 2);
    else RX_STEM_FWD(f16_t, 7);
  }
#undef RX_STEM_FWD
  if (stats) *stat_chunks = bps * 4;
  return 1;
}
