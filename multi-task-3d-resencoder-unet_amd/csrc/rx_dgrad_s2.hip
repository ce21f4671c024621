// rx_dgrad_s2.hip -- backward-data of the stride-2 3x3x3 convolution that opens an encoder stage (resblocks.py:71-74 with
// stride 2), 16-bit compute types, 64 dY channels (two 64-byte chunks) -- the 32 -> 64 layer at 128^3 -> 64^3, whose data
// gradient is the largest launch of the generic gather kernel (igemm<256,32>: 8 parity phases, one 64-byte K step = 2 MFMAs
// per wave between two barriers; 297 us isolated, 195 TF/s).
//
//   dx[2q + r] = sum_{t : (r + 1 - t) even} W[t]^T dy[q + (r + 1 - t)/2]        per axis:   r = 0: t = 1 (offset 0)
//                                                                                              r = 1: t = 0 (offset +1), t = 2 (0)
// so each of the 8 output parity classes is a stride-1 convolution over the dY grid with 1, 2, 4 or 8 taps (27 in total)
// whose offsets are 0 / +1.  A workgroup owns a 4x4x16 tile of dY voxels and 32 output channels: the (5 x 5 x 17)-row dY
// halo of BOTH channel chunks is staged in LDS once and serves all 27 taps of all 8 classes (the gather kernel re-reads a
// dY row from L2 for every tap: 27/8 times per output voxel); per (class, chunk) phase the <= 8 weight slices
// [tap][32 ci][64 B] are staged next to it (next phase's slices wait in registers during the MFMAs), accumulators are 2
// voxel blocks per wave, and each class ends with its own epilogue: 8-byte stores to the voxels (2z+rz, 2y+ry, 2x+rx).
// 71 KB of LDS -> two workgroups per CU.
#include "rx_common.h"

#define DS2_HZ 5
#define DS2_HY 5
#define DS2_HX 17
#define DS2_HV (DS2_HZ * DS2_HY * DS2_HX)   // 425 rows per chunk
#define DS2_XP ((DS2_HV * 4 + 255) / 256)   // 16-byte halo pieces per thread and chunk (7)

struct DgradS2Geom {
  int N, Zo, Yo, Xo;       // dY grid
  int Z, Y, X;             // dX grid (= 2 * dY grid)
  int Ci, Co;              // dX channels (conv input), dY channels (conv output, == 64)
  int ldy, ldx;
  long dy_ss, dx_ss;
  int tz_n, ty_n, tx_n, NT;
  int accumulate, dbg;
};

__device__ inline int ds2_swz(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }
__device__ inline int ds2_lane_voxel(int l) { return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l; }

template <typename T>
__global__ __launch_bounds__(256, 2) void dgrad_s2_halo_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx,
                                                               const DgradS2Geom g) {
  constexpr int P = Elem<T>::PER16;       // 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* sX = reinterpret_cast<u32x4*>(smem);                   // [2 chunks][432 rows][4 pieces]
  u32x4* sW = sX + 2 * 432 * 4;                                 // [8 taps][32 rows][4 pieces]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = rx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int tile = lid / gridDim.y, c0 = (lid - tile * gridDim.y) * 32;       // 32 dX channels
  int tx, ty, tz, n;
  rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, 1, n, tz, ty, tx);
  const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;

  // ---- stage the dY halo of both chunks (rows outside the grid are zero)
  {
    const int chunk = tid & 3;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int p = 0; p < DS2_XP; ++p) {
        const int row = (tid >> 2) + 64 * p;
        if (row < DS2_HV) {
          const int hx = row % DS2_HX, t = row / DS2_HX, hy = t % DS2_HY, hz = t / DS2_HY;
          const int z = z0 + hz, y = y0 + hy, x = x0 + hx;
          u32x4 v = u32x4{0u, 0u, 0u, 0u};
          if (z < g.Zo && y < g.Yo && x < g.Xo)
            v = *reinterpret_cast<const u32x4*>(dy + (long)n * g.dy_ss + ((long)(z * g.Yo + y) * g.Xo + x) * g.ldy + cc * 32 + chunk * P);
          sX[cc * 432 * 4 + ds2_swz(row, chunk)] = v;
        }
      }
  }
  // ---- weight slices of a (class, chunk) phase: pieces ((tl*32 + r)*4 + c4), <= 8*32*4 = 1024 -> 4 per thread
  u32x4 wr[2][4];                    // two phases in flight: a phase is only ~14 MFMAs per wave, far less than a memory round trip
  auto tap_of = [&](int cls, int tl, int& k, int& off) {
    // per axis: parity 0 -> one option (t = 1, offset 0); parity 1 -> two options: (t = 0, +1), (t = 2, 0)
    const int rz = cls >> 2, ry = (cls >> 1) & 1, rx = cls & 1;
    const int nx = rx + 1, ny = ry + 1;
    const int ix = tl % nx, iy = (tl / nx) % ny, iz = tl / (nx * ny);
    const int kz = rz ? 2 * iz : 1, ky = ry ? 2 * iy : 1, kx = rx ? 2 * ix : 1;
    const int oz = rz ? 1 - iz : 0, oy = ry ? 1 - iy : 0, ox = rx ? 1 - ix : 0;
    k = (kz * 3 + ky) * 3 + kx;
    off = (oz * DS2_HY + oy) * DS2_HX + ox;
  };
  auto ntaps_of = [&](int cls) { return ((cls >> 2) + 1) * (((cls >> 1) & 1) + 1) * ((cls & 1) + 1); };
  auto prefetch_w = [&](int ph, int slot) {
    const int cls = ph >> 1, cc = ph & 1, nt = ntaps_of(cls);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = tid + 256 * p;
      const int c4 = i & 3, r = (i >> 2) & 31, tl = i >> 7;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (tl < nt) {
        int k, off;
        tap_of(cls, tl, k, off);
        v = *reinterpret_cast<const u32x4*>(w + ((long)k * g.Ci + c0 + r) * g.Co + cc * 32 + c4 * P);
      }
      wr[slot][p] = v;
    }
  };
  auto commit_w = [&](int slot) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = tid + 256 * p;
      sW[ds2_swz(i >> 2, i & 3)] = wr[slot][p];
    }
  };

  const int fr = lane & 31, fh = lane >> 5;
  const int fv = ds2_lane_voxel(fr);
  int hrow[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int v = (wave * 2 + b) * 32 + fv;
    hrow[b] = (((v >> 6) * DS2_HY) + ((v >> 4) & 3)) * DS2_HX + (v & 15);
  }
  prefetch_w(0, 0);
  prefetch_w(1, 1);
  f32x16 acc[2];
#pragma unroll 2
  for (int ph = 0; ph < 16; ++ph) {
    const int cls = ph >> 1, cc = ph & 1, nt = ntaps_of(cls);
    __syncthreads();                 // previous phase consumed (ph == 0: the halo stores are complete)
    commit_w(ph & 1);
    __syncthreads();
    if (ph + 2 < 16) prefetch_w(ph + 2, ph & 1);
    if (cc == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    }
    const u32x4* sXc = sX + cc * 432 * 4;
    for (int tl = 0; tl < nt; ++tl) {
      int k, off;
      tap_of(cls, tl, k, off);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 af = sW[ds2_swz(tl * 32 + fr, ks * 2 + fh)];
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::run(acc[b], af, sXc[ds2_swz(hrow[b] + off, ks * 2 + fh)]);
      }
    }
    if (cc == 1) {
      // (writing the two x-parities of a 128-byte line together from a second accumulator set was measured SLOWER:
      // 234 vs 192 us -- the half-line stores are not what limits this kernel)
      const int rz = cls >> 2, ry = (cls >> 1) & 1, rx = cls & 1;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int v = (wave * 2 + b) * 32 + fv;
        const int qz = z0 + (v >> 6), qy = y0 + ((v >> 4) & 3), qx = x0 + (v & 15);
        if (qz >= g.Zo || qy >= g.Yo || qx >= g.Xo) continue;
        const int z = 2 * qz + rz, y = 2 * qy + ry, x = 2 * qx + rx;
        T* op = dx + (long)n * g.dx_ss + ((long)(z * g.Y + y) * g.X + x) * g.ldx + c0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int co = 8 * g4 + 4 * fh;
          T vals[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float f = acc[b][4 * g4 + i];
            if (g.accumulate) f += Elem<T>::to_f(op[co + i]);
            vals[i] = Elem<T>::from_f(f);
          }
          *reinterpret_cast<u32x2*>(op + co) = *reinterpret_cast<u32x2*>(vals);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent, weight-stationary variant.  The kernel above is a chain of 16 small phases per tile (~14 MFMAs per wave between two
// barriers and a weight fetch): 192 us.  All 27 weight slices of a 32-channel block are 108 KB and the dY halo of both chunks
// 53 KB: together 1.1 KB MORE than the 160 KB of LDS -- so the single tap of class (0,0,0) lives in registers as ready-made
// A fragments and the other 26 (104 KB) stay in LDS for the workgroup's whole life; one workgroup per CU walks a contiguous
// range of tiles, the next tile's halo waits in registers while the current tile's 216 MFMAs per wave run barrier-free
// (class structure, tap offsets and weight slots are compile-time).  290 us (gather kernel) -> 192 us (kernel above) -> 140 us;
// -0.35 ms per cfg2 step.  Ablation (-DRX_ABLATION=1, RX_DBG): without halo loads 128 us, without MFMAs 144, without stores
// 111, with none of the three 99 us -- the floor is the LDS operand traffic (every one of the 8 waves reads every weight
// fragment: 1.7 MB per tile) plus the weight staging.  Tried: one parity class per wave with its weights as register-resident
// A fragments (LDS then holds only the halo): 128 fragment registers + halo prefetch + accumulators do not fit 256 VGPRs (the
// compiler spilled ~200 registers, 271 us) -- it needs the weight fragments in AGPRs by hand, not attempted.
// ---------------------------------------------------------------------------------------------------------------------
__device__ constexpr int DS2_SLOT_K[26] = {12, 14, 10, 16, 9, 11, 15, 17, 4, 22, 3, 5, 21, 23, 1, 7, 19, 25, 0, 2, 6, 8, 18, 20, 24, 26};
__device__ constexpr int DS2_SLOT_OFF[26] = {1, 0, 17, 0, 18, 17, 1, 0, 85, 0, 86, 85, 1, 0, 102, 85, 17, 0, 103, 102, 86, 85, 18, 17, 1, 0};
__device__ constexpr int DS2_CLS_BASE[8] = {0, 0, 2, 4, 8, 10, 14, 18};
__device__ constexpr int DS2_CLS_NT[8] = {1, 2, 2, 4, 2, 4, 4, 8};
#define DS2P_W_BYTES (26 * 2 * 32 * 64)       // 106,496
#define DS2P_X_BYTES (2 * DS2_HV * 64)        // 54,400
#define DS2P_HP ((2 * DS2_HV * 4 + 511) / 512)  // halo pieces per thread: 7

template <typename T, bool ACC>
__global__ __launch_bounds__(512, 1) void dgrad_s2p_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx,
                                                           const DgradS2Geom g, int tiles_per_wg) {
  constexpr int P = Elem<T>::PER16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;                          // [26 slots][2 chunks][32 rows][64 B], piece XOR (row>>2)&3
  unsigned char* sX = smem + DS2P_W_BYTES;           // [2 chunks][425 rows][64 B], same swizzle
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vb = rx_xcd_remap(blockIdx.x, gridDim.x);
  const int c0 = blockIdx.y * 32;
  const int t_begin = vb * tiles_per_wg, t_end = min(g.NT, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- weights: 26 slices into LDS, the centre tap (class 0) into registers
  for (int i = tid; i < 26 * 2 * 32 * 4; i += 512) {
    const int c4 = i & 3, r = (i >> 2) & 31, cc = (i >> 7) & 1, slot = i >> 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((long)DS2_SLOT_K[slot] * g.Ci + c0 + r) * g.Co + cc * 32 + c4 * P);
    const int row = (slot * 2 + cc) * 32 + r;
    *reinterpret_cast<u32x4*>(sW + row * 64 + ((c4 ^ ((row >> 2) & 3)) << 4)) = v;
  }
  u32x4 wc0[4];                                      // [cc*2 + ks]
#pragma unroll
  for (int j = 0; j < 4; ++j)
    wc0[j] = *reinterpret_cast<const u32x4*>(w + ((long)13 * g.Ci + c0 + fr) * g.Co + (j >> 1) * 32 + ((j & 1) * 2 + fh) * P);

  // ---- halo pieces of this thread: (chunk-of-64B cc, row, piece) decoded once
  int hcoord[DS2P_HP], hdst[DS2P_HP];
#pragma unroll
  for (int p = 0; p < DS2P_HP; ++p) {
    const int i = tid + 512 * p;
    hcoord[p] = -1, hdst[p] = 0;
    if (i < 2 * DS2_HV * 4) {
      const int cc = i / (DS2_HV * 4), rem = i - cc * (DS2_HV * 4), row = rem >> 2, c4 = rem & 3;
      const int hx = row % DS2_HX, t = row / DS2_HX, hy = t % DS2_HY, hz = t / DS2_HY;
      hcoord[p] = (hz << 16) | (hy << 8) | hx;
      hdst[p] = (cc * DS2_HV + row) * 64 + ((c4 ^ ((row >> 2) & 3)) << 4) | (cc << 30) | (c4 << 28);
    }
  }
  u32x4 hq[DS2P_HP];
  auto prefetch_halo = [&](int tile) {
    int tx, ty, tz, n;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, 1, n, tz, ty, tx);
    const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
    const T* dyn = dy + (long)n * g.dy_ss;
#pragma unroll
    for (int p = 0; p < DS2P_HP; ++p) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (hcoord[p] >= 0) {
        const int z = z0 + (hcoord[p] >> 16), y = y0 + ((hcoord[p] >> 8) & 255), x = x0 + (hcoord[p] & 255);
        const int cc = (hdst[p] >> 30) & 1, c4 = (hdst[p] >> 28) & 3;
        if (z < g.Zo && y < g.Yo && x < g.Xo && !RX_ABLATE(g, 1)) v = *reinterpret_cast<const u32x4*>(dyn + ((long)(z * g.Yo + y) * g.Xo + x) * g.ldy + cc * 32 + c4 * P);
      }
      hq[p] = v;
    }
  };
  const int fv = ds2_lane_voxel(fr);
  // 8 waves, one 32-voxel block each: two waves per SIMD hide each other's LDS latency (4 waves x 2 blocks: 145 us)
  constexpr int NB = 1;
  int hrow[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int v = (wave * NB + b) * 32 + fv;
    hrow[b] = (((v >> 6) * DS2_HY) + ((v >> 4) & 3)) * DS2_HX + (v & 15);
  }
  const int wsw = (fr >> 2) & 3;
  prefetch_halo(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                                 // the previous tile's operand reads are done (first tile: nothing)
#pragma unroll
    for (int p = 0; p < DS2P_HP; ++p)
      if (hcoord[p] >= 0) *reinterpret_cast<u32x4*>(sX + (hdst[p] & 0x0fffffff)) = hq[p];
    __syncthreads();                                 // (also publishes the weights before the first tile)
    if (tile + 1 < t_end) prefetch_halo(tile + 1);
    int tx, ty, tz, n;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, 1, n, tz, ty, tx);
    const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
    T* dxn = dx + (long)n * g.dx_ss + c0 + 4 * fh;
#pragma unroll
    for (int cls = 0; cls < 8; ++cls) {
      const int rz = cls >> 2, ry = (cls >> 1) & 1, rx = cls & 1;
      // output addresses of this lane's two voxels for this class (and, ACC, their old values: fetched before the MFMAs)
      T* op[NB];
      bool ok[NB];
      u32x2 oldv[NB][4];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int v = (wave * NB + b) * 32 + fv;
        const int qz = z0 + (v >> 6), qy = y0 + ((v >> 4) & 3), qx = x0 + (v & 15);
        ok[b] = qz < g.Zo && qy < g.Yo && qx < g.Xo;
        op[b] = dxn + ((long)((2 * qz + rz) * g.Y + 2 * qy + ry) * g.X + 2 * qx + rx) * g.ldx;
        if (ACC) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) oldv[b][g4] = ok[b] ? *reinterpret_cast<const u32x2*>(op[b] + 8 * g4) : u32x2{0u, 0u};
        }
      }
      f32x16 acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      __builtin_amdgcn_sched_barrier(0);             // keep the 8 classes apart: left alone the scheduler hoists every fragment read
#pragma unroll                                       // of the tile (512 registers + spills)
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int tl = 0; tl < DS2_CLS_NT[cls]; ++tl) {
          const int slot = DS2_CLS_BASE[cls] + tl;
          const int off = cls == 0 ? 0 : DS2_SLOT_OFF[slot];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            u32x4 af;
            if (cls == 0)
              af = wc0[cc * 2 + ks];
            else
              af = *reinterpret_cast<const u32x4*>(sW + ((slot * 2 + cc) * 32 + fr) * 64 + (((ks * 2 + fh) ^ wsw) << 4));
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const int row = hrow[b] + off;
              const u32x4 bf = *reinterpret_cast<const u32x4*>(sX + (cc * DS2_HV + row) * 64 + (((ks * 2 + fh) ^ ((row >> 2) & 3)) << 4));
              if (!RX_ABLATE(g, 4)) Mma<T>::run(acc[b], af, bf);
              else acc[b][0] += __builtin_bit_cast(float, af[0] ^ bf[0]);
            }
          }
        }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!ok[b] || RX_ABLATE(g, 8)) continue;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          T vals[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float f = acc[b][4 * g4 + i];
            if (ACC) f += Elem<T>::to_f(reinterpret_cast<const T*>(&oldv[b][g4])[i]);
            vals[i] = Elem<T>::from_f(f);
          }
          *reinterpret_cast<u32x2*>(op[b] + 8 * g4) = *reinterpret_cast<u32x2*>(vals);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// returns 1 if handled, 0 to fall through to the generic kernel, negative on error
int rx_dgrad_s2_halo_try(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx, int accumulate, hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("RX_DGRAD_S2");
    on = e ? atoi(e) : 1;
  }
  if (!on || dt == RX_F32 || dy->c != 64 || dx->c % 32 || dy->ld % 8 || dx->ld % 4 || ((uintptr_t)dy->ptr & 15) || ((uintptr_t)dx->ptr & 7) ||
      ((uintptr_t)w_bwd & 15))
    return 0;
  if (dx->z != 2 * dy->z || dx->y != 2 * dy->y || dx->x != 2 * dy->x) return 0;     // even input extents only
  if (rx_act_voxels(dx) * (long)dx->ld >= (1L << 31) || rx_act_voxels(dy) * (long)dy->ld >= (1L << 31)) return 0;
  DgradS2Geom g;
  memset(&g, 0, sizeof(g));
  g.N = dy->n, g.Zo = dy->z, g.Yo = dy->y, g.Xo = dy->x;
  g.Z = dx->z, g.Y = dx->y, g.X = dx->x;
  g.Ci = dx->c, g.Co = dy->c, g.ldy = dy->ld, g.ldx = dx->ld;
  g.dy_ss = rx_act_voxels(dy) * (long)dy->ld, g.dx_ss = rx_act_voxels(dx) * (long)dx->ld;
  g.tz_n = (g.Zo + 3) / 4, g.ty_n = (g.Yo + 3) / 4, g.tx_n = (g.Xo + 15) / 16;
  g.NT = g.N * g.tz_n * g.ty_n * g.tx_n;
  g.accumulate = accumulate;
  { const char* e = getenv("RX_DBG"); g.dbg = e ? atoi(e) : 0; }
  if ((long)g.NT * (g.Ci / 32) < 256) return 0;                 // small layers: the gather kernel's split-K fills the chip better
  const size_t lds = (size_t)(2 * 432 * 4 + 8 * 32 * 4) * 16;   // 71,680 B
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2_halo_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2_halo_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  static int pers = -1;
  if (pers < 0) {
    const char* e = getenv("RX_DGRAD_S2P");
    pers = e ? atoi(e) : 1;
  }
  if (pers) {
    const size_t ldsp = (size_t)DS2P_W_BYTES + DS2P_X_BYTES;      // 160,896 B
    static bool attrp = false;
    if (!attrp) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2p_kernel<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2p_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2p_kernel<f16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_s2p_kernel<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      attrp = true;
    }
    int wgs = 256 / (g.Ci / 32);
    if (wgs < 1) wgs = 1;
    if (wgs > g.NT) wgs = g.NT;
    const int per = (g.NT + wgs - 1) / wgs;
    wgs = (g.NT + per - 1) / per;
    rx_note_kernel("dgrad_s2p_kernel");
    dim3 gridp(wgs, g.Ci / 32);
#define RX_DS2P(TT, A) hipLaunchKernelGGL((dgrad_s2p_kernel<TT, A>), gridp, dim3(512), ldsp, st, (const TT*)dy->ptr, (const TT*)w_bwd, (TT*)dx->ptr, g, per)
    if (dt == RX_BF16) {
      if (accumulate) RX_DS2P(bf16_t, true); else RX_DS2P(bf16_t, false);
    } else {
      if (accumulate) RX_DS2P(f16_t, true); else RX_DS2P(f16_t, false);
    }
#undef RX_DS2P
    hipError_t ep = hipGetLastError();
    if (ep != hipSuccess) {
      rx_set_error("dgrad_s2p: %s", hipGetErrorString(ep));
      return RX_ELAUNCH;
    }
    return 1;
  }
  rx_note_kernel("dgrad_s2_halo_kernel");
  dim3 grid(g.NT, g.Ci / 32);
  if (dt == RX_BF16)
    hipLaunchKernelGGL((dgrad_s2_halo_kernel<bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)dy->ptr, (const bf16_t*)w_bwd, (bf16_t*)dx->ptr, g);
  else
    hipLaunchKernelGGL((dgrad_s2_halo_kernel<f16_t>), grid, dim3(256), lds, st, (const f16_t*)dy->ptr, (const f16_t*)w_bwd, (f16_t*)dx->ptr, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rx_set_error("dgrad_s2_halo: %s", hipGetErrorString(e));
    return RX_ELAUNCH;
  }
  return 1;
}
