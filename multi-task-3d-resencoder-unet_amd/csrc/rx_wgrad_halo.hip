// rx_wgrad_halo.hip -- weight gradient of the stride-1 3x3x3 convolutions (92 % of the conv FLOPs)
// with an LDS halo tile, 16-bit types.
//
//   dW[t][co][ci] = sum_v dY[v][co] * X[v + d_t][ci]
//
// The generic kernel (rx_wgrad.hip) launches one workgroup per tap, so every X / dY voxel is fetched
// from L2 27 times.  Here ONE workgroup owns a (32 co) x (32 ci) panel pair and ALL 27 taps:
//   * per spatial tile (TZ x TY x TX <= 256 voxels) it stages the dY panel [VT][32] and the X halo panel
//     [(TZ+2)(TY+2)(TX+2)][32] in LDS once (64-byte rows, zero outside the volume);
//   * the 4 waves split the taps (7/7/7/6); per 16-voxel k-step a wave fetches the dY fragment once and one
//     X fragment per tap -- the tap shift is just a row offset into the halo panel -- with
//     ds_read_b64_tr_b16 (4 consecutive voxels x 16 channels per 16-lane group: any 4 consecutive 64-byte
//     rows span all 64 banks exactly once -> conflict-free for every tap);
//   * 7 x 32x32 fp32 accumulators per wave live in registers across ALL tiles of the workgroup's split and
//     are written once as slab[split][tap][R][C]; the fixed-order reduce of rx_wgrad.hip finishes.
// Two workgroups per CU (<= 62 KB LDS each) overlap one's staging with the other's MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "rx_common.h"

struct WgHaloGeom {
  int N, Z, Y, X;
  int R, Cc, ldg, ldx;
  long g_ss, x_ss;
  long x_cs;             // planar concat x (rx_act.cs): element offset of 32-channel group j = j * x_cs; 0 = contiguous channels
  int TZ, TY, TX, lTX, lTY;
  int HY, HX, HV, VT;
  int tz_n, ty_n, tx_n, NT;
  int S, tiles_per_split, panels_c;
  int sz, sy, sx;        // conv stride per axis (1 or 2); the X halo of a TZ x TY x TX tile of dY is (s*(T-1)+3) per axis
  int Zi, Yi, Xi;        // extent of X (the conv INPUT); Z, Y, X above are the extent of dY
  int order; // tile walk order (rx_tile_coords)
  int dbg;   // ablation mask (RX_DBG env): 1 stage only the first tile, 2 no MFMA loop
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_h;

#define RX_WGH_MAX_HV 768
#define RX_WGH_MAX_VT 256
#define RX_WGH_XPIECES ((RX_WGH_MAX_HV * 4 + 255) / 256)  // 16-byte pieces of the halo panel per thread
#define RX_WGH_GPIECES (RX_WGH_MAX_VT * 4 / 256)

template <typename T>
__device__ inline u32x4 tr_frag(const T* p0, const T* p1) {
  s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_h*)(p0));
  s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_h*)(p1));
  u32x2 lo = __builtin_bit_cast(u32x2, t0), hi = __builtin_bit_cast(u32x2, t1);
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_halo_kernel(const T* __restrict__ gt, const T* __restrict__ xt, float* __restrict__ slab,
                                                            float* __restrict__ dw, const WgHaloGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sG = reinterpret_cast<T*>(smem);                       // [VT][32]
  T* sX = sG + RX_WGH_MAX_VT * 32;                          // [HV][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: tr reads need full EXEC
  // logical id = split * PP + panel pair: the PP workgroups of one split read the SAME tiles (different channel panels)
  // and neighbouring splits share halos -- keep them on one XCD
  const int PPn = gridDim.x, lid = rx_xcd_remap(blockIdx.y * PPn + blockIdx.x, PPn * gridDim.y);
  const int split = lid / PPn, pp = lid - split * PPn;
  const int pr = pp / g.panels_c, pc = pp - pr * g.panels_c;
  const int r0 = pr * 32, c0 = pc * 32;
  const long xc0 = g.x_cs ? (long)pc * g.x_cs : (long)c0;      // where this panel's 32 input channels start
  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(g.NT, t_begin + g.tiles_per_split);

  // ---- per-thread staging geometry (tile independent)
  const int chunk = tid & 3;                                 // 16-byte chunk of a 64-byte row
  int xh[RX_WGH_XPIECES];                                    // packed halo coordinates (hz<<16 | hy<<8 | hx), -1 = none
#pragma unroll
  for (int p = 0; p < RX_WGH_XPIECES; ++p) {
    int row = (tid >> 2) + 64 * p;
    if (row < g.HV) {
      int hx = row % g.HX, t = row / g.HX;
      int hy = t % g.HY, hz = t / g.HY;
      xh[p] = (hz << 16) | (hy << 8) | hx;
    } else
      xh[p] = -1;
  }

  // ---- per-lane fragment geometry
  const int g16 = lane >> 4, half = g16 & 1, h = g16 >> 1, l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3;
  const int coloff = 16 * half + 4 * p4;                     // element offset inside a 32-channel row
  int toff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    int t = wave + 4 * j;
    int dz = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
    toff[j] = t < 27 ? ((dz * g.HY + dy) * g.HX + dx) * 32 : 0;
  }

  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int ksteps = g.VT >> 4;
  // Staging is software pipelined: the global loads of tile i+1 are issued right after tile i has been committed to
  // LDS and stay in flight (in registers) while tile i's MFMAs run.
  u32x4 gv[RX_WGH_GPIECES], xv[RX_WGH_XPIECES];
  auto load_tile = [&](int tile) {
    // tile -> (n, z0, y0, x0)
    int tx, ty, tz, n;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
    const int z0 = tz * g.TZ, y0 = ty * g.TY, x0 = tx * g.TX;
    const T* gn = gt + n * g.g_ss + r0;
    const T* xn = xt + n * g.x_ss + xc0;
#pragma unroll
    for (int p = 0; p < RX_WGH_GPIECES; ++p) {
      int v = (tid >> 2) + 64 * p;
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (v < g.VT) {
        int vx = v & (g.TX - 1), vy = (v >> g.lTX) & (g.TY - 1), vz = v >> (g.lTX + g.lTY);
        int z = z0 + vz, y = y0 + vy, x = x0 + vx;
        if (z < g.Z && y < g.Y && x < g.X)
          val = *reinterpret_cast<const u32x4*>(gn + ((long)(z * g.Y + y) * g.X + x) * g.ldg + chunk * 8);
      }
      gv[p] = val;
    }
#pragma unroll
    for (int p = 0; p < RX_WGH_XPIECES; ++p) {
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (xh[p] >= 0) {
        int z = z0 * g.sz + (xh[p] >> 16) - 1, y = y0 * g.sy + ((xh[p] >> 8) & 255) - 1, x = x0 * g.sx + (xh[p] & 255) - 1;
        if ((unsigned)z < (unsigned)g.Zi && (unsigned)y < (unsigned)g.Yi && (unsigned)x < (unsigned)g.Xi)
          val = *reinterpret_cast<const u32x4*>(xn + ((long)(z * g.Yi + y) * g.Xi + x) * g.ldx + chunk * 8);
      }
      xv[p] = val;
    }
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
    for (int p = 0; p < RX_WGH_GPIECES; ++p) {
      int v = (tid >> 2) + 64 * p;
      if (v < g.VT) *reinterpret_cast<u32x4*>(sG + v * 32 + chunk * 8) = gv[p];
    }
#pragma unroll
    for (int p = 0; p < RX_WGH_XPIECES; ++p) {
      int row = (tid >> 2) + 64 * p;
      if (xh[p] >= 0) *reinterpret_cast<u32x4*>(sX + row * 32 + chunk * 8) = xv[p];
    }
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
    // ---- compute: 16-voxel k-steps; this lane supplies rows (voxels) v, v+4 of its 8-voxel half
    for (int s = 0; s < ksteps; ++s) {
      const int v = 16 * s + 8 * h + q4;
      const int vx = v & (g.TX - 1), vy = (v >> g.lTX) & (g.TY - 1), vz = v >> (g.lTX + g.lTY);
      // halo row of the CENTRE tap of output voxel v (input voxel s*v); v+4 is 4*sx rows further when TX >= 8
      const int hr = ((vz * g.sz + 1) * g.HY + (vy * g.sy + 1)) * g.HX + vx * g.sx + 1;
      int hr4;
      if (g.TX >= 8)
        hr4 = hr + 4 * g.sx;
      else {  // TX == 4: v+4 is the next x-row
        const int v2 = v + 4;
        const int vy2 = (v2 >> g.lTX) & (g.TY - 1), vz2 = v2 >> (g.lTX + g.lTY);
        hr4 = ((vz2 * g.sz + 1) * g.HY + (vy2 * g.sy + 1)) * g.HX + (v2 & (g.TX - 1)) * g.sx + 1;
      }
      const u32x4 a = tr_frag<T>(sG + v * 32 + coloff, sG + (v + 4) * 32 + coloff);
      const T* b0 = sX + hr * 32 + coloff;
      const T* b1 = sX + hr4 * 32 + coloff;
      // all fragment reads of the k-step are issued before its first MFMA (distinct registers): the LDS latency is
      // paid once per 7 MFMAs instead of once per MFMA (the compiler otherwise recycles ONE fragment register set and
      // brackets every MFMA with s_waitcnt lgkmcnt(0))
      u32x4 bf[7];
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (j < 6 || wave < 3) bf[j] = tr_frag<T>(b0 + toff[j], b1 + toff[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (j < 6 || wave < 3) Mma<T>::run(acc[j], a, bf[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const int col = lane & 31, fh = lane >> 5;
  if (g.S == 1) {
    // ---- single split (low-resolution layers: many panel pairs, few tiles): no slab, no reduce launch.  The panel is
    // transposed through LDS 16 rows at a time ([16][32][27] floats = 54 KB) and leaves as contiguous 3456-byte runs of
    // dw[r][c0 .. c0+31][0..26]  (tap stride 27 floats is odd -> the 32 columns of a store hit 32 distinct banks).
    float* sT = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      __syncthreads();  // tile loop / previous half fully drained
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int t = wave + 4 * j;
        if (t < 27) {
#pragma unroll
          for (int r = 8 * hb; r < 8 * hb + 8; ++r) {
            const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * fh;  // row within this half
            sT[(lr * 32 + col) * 27 + t] = acc[j][r];
          }
        }
      }
      __syncthreads();
      for (int idx = tid; idx < 16 * 216; idx += 256) {   // float4 pieces: 216 per 32-column row run (3456 B, 16-B aligned)
        const int lr = idx / 216, i = idx - lr * 216;
        *reinterpret_cast<f32x4*>(dw + ((long)(r0 + 16 * hb + lr) * g.Cc + c0) * 27 + 4 * i) = *reinterpret_cast<const f32x4*>(sT + 4 * idx);
      }
    }
    return;
  }
  // ---- write the 7 accumulators: slab[split][tap][r0+row][c0+col]
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int t = wave + 4 * j;
    if (t < 27) {
      float* out = slab + (((long)split * 27 + t) * g.R + r0) * g.Cc + c0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
        out[(long)row * g.Cc + col] = acc[j][r];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Specialisation for the 4 x 4 x 16 tile (every layer whose X extent is >= 16: 92 % of the weight-gradient FLOPs).
// SQ counters of the generic kernel above: 9.3 VALU instructions per MFMA (voxel decode and one address add per
// fragment read) -- with two waves per SIMD the VALU, not the matrix pipe (38 % busy), was the bound.  Here the tile
// geometry AND the wave's tap set are compile-time constants (one instantiation per wave index, chosen by a scalar
// branch), the 16 k-steps are unrolled, and every fragment address is  lane_base + IMMEDIATE  in the ds_read offset
// field: the MFMA loop contains no address arithmetic at all.  Fragment reads are double buffered across k-steps.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ inline u32x4 tr_frag_at(const lds_byte* p, int off0, int off1) {
  s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_h*)(p + off0));
  s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_h*)(p + off1));
  u32x2 lo = __builtin_bit_cast(u32x2, t0), hi = __builtin_bit_cast(u32x2, t1);
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

#define WGH16_HY 6
#define WGH16_HX 18
#define WGH16_HV 648
#define WGH16_XPIECES 11   // ceil(648 * 4 / 256)

// gb: lane base into sG (row 8h+q4 of k-step 0), xb: lane base into sX (halo row (0,0,8h+q4) i.e. tap (-1,-1,-1) of
// voxel (0,0,8h+q4)); both include the lane's column offset.  Rows are 64 bytes.  `stage(s)` is called once per k-step
// between the MFMAs: the caller uses it to issue one piece of the NEXT tile's global loads, so that their address
// arithmetic (VALU) and latency disappear behind the matrix pipe.
template <typename T, int W, typename Stage>
__device__ __forceinline__ void wgh16_tile_mma(const lds_byte* gb, const lds_byte* xb, f32x16 (&acc)[7], Stage&& stage) {
  constexpr int NT = W < 3 ? 7 : 6;
  // Rolling single buffer: the fragment of tap j for k-step s+1 is read into the registers of tap j right after the
  // MFMA of (s, j) has issued, i.e. 7 MFMAs (~220 cycles) before its consumer -- the whole LDS latency is hidden with
  // one B register set; only the A fragment (shared by the 7 MFMAs of a step) is double buffered.
  u32x4 fa[2], fb[7];
  auto read_a = [&](int s) { return tr_frag_at(gb, s * 1024, s * 1024 + 256); };
  auto read_b = [&](int s, int j) {
    const int vy = s & 3, vz = s >> 2;
    const int t = W + 4 * j;
    const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;                   // 0..2 (the lane base sits at tap (-1,-1,-1))
    const int off = (((vz + dz) * WGH16_HY + (vy + dy)) * WGH16_HX + dx) * 64;
    return tr_frag_at(xb, off, off + 256);
  };
  fa[0] = read_a(0);
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[j] = read_b(0, j);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s + 1 < 16) fa[(s + 1) & 1] = read_a(s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      Mma<T>::run(acc[j], fa[s & 1], fb[j]);
      if (s + 1 < 16) fb[j] = read_b(s + 1, j);
      __builtin_amdgcn_sched_barrier(0);   // pin the pair: the scheduler otherwise sinks every read to just before its
                                           // consumer, recycles ONE register set and brackets each MFMA with lgkmcnt(0)
    }
    stage(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

#define WGH16_BUF_BYTES ((256 + WGH16_HV) * 64)   // one (dY tile, X halo tile) pair: 57,856 bytes

// One workgroup per CU (the pipelined loop wants ~370 registers per lane).  Measured and rejected: a second LDS tile buffer
// (one barrier per tile) and issuing the next tile's loads piecewise between the MFMAs -- 4-14 % slower in isolation, and
// a 115 KB workgroup no longer shares a CU with the main stream's convolutions (+1.2 ms per cfg2 step).
template <typename T>
__global__ __launch_bounds__(256, 1) void wgrad_halo16_kernel(const T* __restrict__ gt, const T* __restrict__ xt, float* __restrict__ slab,
                                                              float* __restrict__ dw, const WgHaloGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // sG [256][32], sX [648][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int PPn = gridDim.x, lid = rx_xcd_remap(blockIdx.y * PPn + blockIdx.x, PPn * gridDim.y);
  const int split = lid / PPn, pp = lid - split * PPn;
  const int pr = pp / g.panels_c, pc = pp - pr * g.panels_c;
  const int r0 = pr * 32, c0 = pc * 32;
  const long xc0 = g.x_cs ? (long)pc * g.x_cs : (long)c0;      // where this panel's 32 input channels start
  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(g.NT, t_begin + g.tiles_per_split);

  const int chunk = tid & 3;
  int xh[WGH16_XPIECES];                                     // packed halo coordinates (hz<<16 | hy<<8 | hx), -1 = none
#pragma unroll
  for (int p = 0; p < WGH16_XPIECES; ++p) {
    const int row = (tid >> 2) + 64 * p;
    const int hx = row % WGH16_HX, t = row / WGH16_HX;
    xh[p] = row < WGH16_HV ? ((t / WGH16_HY) << 16) | ((t % WGH16_HY) << 8) | hx : -1;
  }
  const int g16 = lane >> 4, half = g16 & 1, h = g16 >> 1, l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3;
  const int lane_off = ((8 * h + q4) * 32 + 16 * half + 4 * p4) * 2;   // bytes: row 8h+q4, this lane's 4-channel column group

  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- staging: 15 sixteen-byte pieces per thread and tile (4 of dY, 11 of the X halo)
  u32x4 gv[4], xv[WGH16_XPIECES];
  int s_z0 = 0, s_y0 = 0, s_x0 = 0;
  const T* s_gn = gt;
  const T* s_xn = xt;
  auto set_tile = [&](int tile) {       // wave-uniform decode of the tile being staged
    int tx, ty, tz, n;
    rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
    s_z0 = tz * 4, s_y0 = ty * 4, s_x0 = tx * 16;
    s_gn = gt + n * g.g_ss + r0 + chunk * 8;
    s_xn = xt + n * g.x_ss + xc0 + chunk * 8;
  };
  auto issue_piece = [&](int p) {       // p in [0, 15); p is a compile-time constant at every call site
    if (p < 4) {
      const int v = (tid >> 2) + 64 * p;
      const int z = s_z0 + (v >> 6), y = s_y0 + ((v >> 4) & 3), x = s_x0 + (v & 15);
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (z < g.Z && y < g.Y && x < g.X) val = *reinterpret_cast<const u32x4*>(s_gn + ((long)(z * g.Y + y) * g.X + x) * g.ldg);
      gv[p] = val;
    } else {
      const int q = p - 4;
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (xh[q] >= 0) {
        const int z = s_z0 + (xh[q] >> 16) - 1, y = s_y0 + ((xh[q] >> 8) & 255) - 1, x = s_x0 + (xh[q] & 255) - 1;
        if ((unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X)
          val = *reinterpret_cast<const u32x4*>(s_xn + ((long)(z * g.Y + y) * g.X + x) * g.ldx);
      }
      xv[q] = val;
    }
  };
  auto commit = [&](int buf) {          // staged registers -> LDS buffer `buf`
    T* sG = reinterpret_cast<T*>(smem + buf * WGH16_BUF_BYTES);
    T* sX = sG + 256 * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(sG + ((tid >> 2) + 64 * p) * 32 + chunk * 8) = gv[p];
#pragma unroll
    for (int p = 0; p < WGH16_XPIECES; ++p)
      if (xh[p] >= 0) *reinterpret_cast<u32x4*>(sX + ((tid >> 2) + 64 * p) * 32 + chunk * 8) = xv[p];
  };

  if (t_begin < t_end) {
    set_tile(t_begin);
#pragma unroll
    for (int p = 0; p < 15; ++p) issue_piece(p);
  }
  const lds_byte* gb = (const lds_byte*)(smem) + lane_off;
  const lds_byte* xb = gb + 256 * 64;
  auto stage = [](int) {};
  // The WHOLE tile loop sits inside the per-wave arm: with the switch inside the loop the accumulators crossed a phi
  // at every iteration and were copied AGPR <-> VGPR once per tile (112 v_accvgpr moves = 1 VALU op per MFMA).
  auto run = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    for (int tile = t_begin; tile < t_end; ++tile) {
      if (!RX_ABLATE(g, 1) || tile == t_begin) {
        __syncthreads();          // every wave is done reading the previous tile
        commit(0);
        __syncthreads();
        if (tile + 1 < t_end) {   // the next tile's loads stay in flight (in registers) behind this tile's MFMAs
          set_tile(tile + 1);
#pragma unroll
          for (int p = 0; p < 15; ++p) issue_piece(p);
        }
      }
      if (!RX_ABLATE(g, 2)) wgh16_tile_mma<T, W>(gb, xb, acc, stage);
    }
  };
  switch (wave) {   // scalar branch: the tap set of a wave is a compile-time constant inside each arm
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }

  const int col = lane & 31, fh = lane >> 5;
  if (g.S == 1) {   // single split: transpose through LDS, contiguous runs of dw (see the generic kernel)
    float* sT = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int t = wave + 4 * j;
        if (t < 27) {
#pragma unroll
          for (int r = 8 * hb; r < 8 * hb + 8; ++r) {
            const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * fh;
            sT[(lr * 32 + col) * 27 + t] = acc[j][r];
          }
        }
      }
      __syncthreads();
      for (int idx = tid; idx < 16 * 216; idx += 256) {
        const int lr = idx / 216, i = idx - lr * 216;
        *reinterpret_cast<f32x4*>(dw + ((long)(r0 + 16 * hb + lr) * g.Cc + c0) * 27 + 4 * i) = *reinterpret_cast<const f32x4*>(sT + 4 * idx);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int t = wave + 4 * j;
    if (t < 27) {
      float* out = slab + (((long)split * 27 + t) * g.R + r0) * g.Cc + c0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
        out[(long)row * g.Cc + col] = acc[j][r];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant of the 4x4x16 kernel: 512 threads = 4 CONSUMER waves (the MFMA loop above, 7 taps each, no
// staging registers) + 4 PRODUCER waves (global -> registers -> LDS for the next tile), one of each per SIMD, over two
// LDS tile buffers and ONE barrier per tile.  Ablation of the 256-thread kernel: MFMA loop alone 221 us, staging alone
// 193 us, together 343 us (32->32 @128^3): in a single instruction stream the two only overlap by the load latency.
// Here the producers' address arithmetic, load waits and ds_writes run beside the consumers' MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
// DMA = true: the producers move both tiles global -> LDS with buffer_load_dwordx4 ... lds (1 KiB per wave-instruction,
// lane-linear: exactly the [row][64 B] image this kernel uses), no staging registers, no ds_write pass; voxels outside the
// volume get an out-of-range buffer offset and the hardware writes ZEROS for them (probed: scripts/probes/buffer_lds_oob.hip).
template <typename T, bool DMA>
__global__ __launch_bounds__(512, 1) void wgrad_halo16ws_kernel(const T* __restrict__ gt, const T* __restrict__ xt, float* __restrict__ slab,
                                                                float* __restrict__ dw, const WgHaloGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x { sG [256][32], sX [648][32] }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // 0-3 consumers, 4-7 producers
  const int PPn = gridDim.x, lid = rx_xcd_remap(blockIdx.y * PPn + blockIdx.x, PPn * gridDim.y);
  const int split = lid / PPn, pp = lid - split * PPn;
  const int pr = pp / g.panels_c, pc = pp - pr * g.panels_c;
  const int r0 = pr * 32, c0 = pc * 32;
  const long xc0 = g.x_cs ? (long)pc * g.x_cs : (long)c0;      // where this panel's 32 input channels start
  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(g.NT, t_begin + g.tiles_per_split);

  f32x16 acc[7];
  if (wave >= 4) {
    // ================================= producers =================================
    const int ptid = tid - 256, chunk = ptid & 3;
    int xh[WGH16_XPIECES];
#pragma unroll
    for (int p = 0; p < WGH16_XPIECES; ++p) {
      const int row = (ptid >> 2) + 64 * p;
      const int hx = row % WGH16_HX, t = row / WGH16_HX;
      xh[p] = row < WGH16_HV ? ((t / WGH16_HY) << 16) | ((t % WGH16_HY) << 8) | hx : -1;
    }
    u32x4 gv[4], xv[WGH16_XPIECES];
    auto load_tile = [&](int tile) {
      int tx, ty, tz, n;
      rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
      const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
      const T* gn = gt + n * g.g_ss + r0 + chunk * 8;
      const T* xn = xt + n * g.x_ss + xc0 + chunk * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int v = (ptid >> 2) + 64 * p;
        const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (z < g.Z && y < g.Y && x < g.X) val = *reinterpret_cast<const u32x4*>(gn + ((long)(z * g.Y + y) * g.X + x) * g.ldg);
        gv[p] = val;
      }
#pragma unroll
      for (int p = 0; p < WGH16_XPIECES; ++p) {
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (xh[p] >= 0) {
          const int z = z0 + (xh[p] >> 16) - 1, y = y0 + ((xh[p] >> 8) & 255) - 1, x = x0 + (xh[p] & 255) - 1;
          if ((unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X)
            val = *reinterpret_cast<const u32x4*>(xn + ((long)(z * g.Y + y) * g.X + x) * g.ldx);
        }
        xv[p] = val;
      }
    };
    auto commit = [&](int buf) {
      T* sG = reinterpret_cast<T*>(smem + buf * WGH16_BUF_BYTES);
      T* sX = sG + 256 * 32;
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(sG + ((ptid >> 2) + 64 * p) * 32 + chunk * 8) = gv[p];
#pragma unroll
      for (int p = 0; p < WGH16_XPIECES; ++p)
        if (xh[p] >= 0) *reinterpret_cast<u32x4*>(sX + ((ptid >> 2) + 64 * p) * 32 + chunk * 8) = xv[p];
    };
    if (DMA) {
      const unsigned OOB = 0x7fffff00u;
      const long gbytes = (long)g.N * g.g_ss * 2, xbytes = ((long)g.N * g.x_ss + (g.x_cs ? (long)(g.Cc / 32 - 1) * g.x_cs : 0)) * 2;
      __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)gt, 0, (unsigned)(gbytes > 0xfffffff0L ? 0xfffffff0L : gbytes), 0x00020000);
      __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)xt, 0, (unsigned)(xbytes > 0xfffffff0L ? 0xfffffff0L : xbytes), 0x00020000);
      const int pw = wave - 4;
      auto dma_tile = [&](int tile, int buf) {
        int tx, ty, tz, n;
        rx_tile_coords(tile, g.tx_n, g.ty_n, g.tz_n, g.order, n, tz, ty, tx);
        const int z0 = tz * 4, y0 = ty * 4, x0 = tx * 16;
        const unsigned gbase = (unsigned)((n * g.g_ss + r0 + chunk * 8) * 2);
        const unsigned xbase = (unsigned)((n * g.x_ss + xc0 + chunk * 8) * 2);
        __attribute__((address_space(3))) unsigned char* lG =
            (__attribute__((address_space(3))) unsigned char*)(smem + buf * WGH16_BUF_BYTES) + 1024 * pw;
        __attribute__((address_space(3))) unsigned char* lX = lG + 256 * 64;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int v = (ptid >> 2) + 64 * p;
          const int z = z0 + (v >> 6), y = y0 + ((v >> 4) & 3), x = x0 + (v & 15);
          const unsigned off = (z < g.Z && y < g.Y && x < g.X) ? gbase + (unsigned)(((z * g.Y + y) * g.X + x) * g.ldg * 2) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (__attribute__((address_space(3))) void*)(lG + 4096 * p), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < WGH16_XPIECES; ++p) {
          if (xh[p] >= 0) {      // rows >= 648 of the last piece: lanes switched off, nothing is written
            const int z = z0 + (xh[p] >> 16) - 1, y = y0 + ((xh[p] >> 8) & 255) - 1, x = x0 + (xh[p] & 255) - 1;
            const bool ok = (unsigned)z < (unsigned)g.Z && (unsigned)y < (unsigned)g.Y && (unsigned)x < (unsigned)g.X;
            const unsigned off = ok ? xbase + (unsigned)(((z * g.Y + y) * g.X + x) * g.ldx * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)(lX + 4096 * p), 16, off, 0, 0, 0);
          }
        }
      };
      if (t_begin < t_end) dma_tile(t_begin, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // tile 0 visible
      for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        if (tile + 1 < t_end) dma_tile(tile + 1, buf ^ 1);   // its readers passed the barrier that ended the previous iteration
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
    if (t_begin < t_end) {
      load_tile(t_begin);
      commit(0);
      if (t_begin + 1 < t_end) load_tile(t_begin + 1);
    }
    __syncthreads();                                   // tile 0 visible
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int buf = (tile - t_begin) & 1;
      if (tile + 1 < t_end) {
        commit(buf ^ 1);                               // its readers passed the barrier that ended the previous iteration
        if (tile + 2 < t_end) load_tile(tile + 2);     // in flight during the next iteration's commit-free time
      }
      __syncthreads();
    }
    }
  } else {
    // ================================= consumers =================================
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int g16 = lane >> 4, half = g16 & 1, h = g16 >> 1, l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3;
    const int lane_off = ((8 * h + q4) * 32 + 16 * half + 4 * p4) * 2;
    auto stage = [](int) {};
    auto run = [&](auto wc) {
      constexpr int W = decltype(wc)::value;
      __syncthreads();                                 // tile 0 visible
      for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        const lds_byte* gb = (const lds_byte*)(smem) + buf * WGH16_BUF_BYTES + lane_off;
        if (!RX_ABLATE(g, 2)) wgh16_tile_mma<T, W>(gb, gb + 256 * 64, acc, stage);
        __syncthreads();
      }
    };
    switch (wave) {
      case 0: run(std::integral_constant<int, 0>{}); break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: run(std::integral_constant<int, 2>{}); break;
      default: run(std::integral_constant<int, 3>{}); break;
    }
  }

  // ---- epilogue: the consumers own the accumulators; every thread helps with the copy-out of the single-split path
  const int col = lane & 31, fh = lane >> 5;
  if (g.S == 1) {
    float* sT = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      __syncthreads();
      if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int t = wave + 4 * j;
          if (t < 27) {
#pragma unroll
            for (int r = 8 * hb; r < 8 * hb + 8; ++r) {
              const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * fh;
              sT[(lr * 32 + col) * 27 + t] = acc[j][r];
            }
          }
        }
      }
      __syncthreads();
      for (int idx = tid; idx < 16 * 216; idx += 512) {
        const int lr = idx / 216, i = idx - lr * 216;
        *reinterpret_cast<f32x4*>(dw + ((long)(r0 + 16 * hb + lr) * g.Cc + c0) * 27 + 4 * i) = *reinterpret_cast<const f32x4*>(sT + 4 * idx);
      }
    }
    return;
  }
  if (wave < 4) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int t = wave + 4 * j;
      if (t < 27) {
        float* out = slab + (((long)split * 27 + t) * g.R + r0) * g.Cc + c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
          out[(long)row * g.Cc + col] = acc[j][r];
        }
      }
    }
  }
}

static int p2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// returns 1 and fills `g` when the halo kernel applies
static int wgh_plan(const rx_act* x, const rx_act* dy, const int32_t stride[3], WgHaloGeom* g, size_t ws_bytes) {
  memset(g, 0, sizeof(*g));
  g->N = dy->n, g->Z = dy->z, g->Y = dy->y, g->X = dy->x;
  g->Zi = x->z, g->Yi = x->y, g->Xi = x->x;
  g->sz = stride[0], g->sy = stride[1], g->sx = stride[2];
  g->R = dy->c, g->Cc = x->c, g->ldg = dy->ld, g->ldx = x->ld;
  g->g_ss = rx_act_voxels(dy) * (long)dy->ld;
  g->x_ss = rx_act_voxels(x) * (long)x->ld;
  g->x_cs = x->cs;
  const bool strided = stride[0] > 1 || stride[1] > 1 || stride[2] > 1;
  int TX = p2ceil(g->X);
  TX = TX < 4 ? 4 : (TX > 16 ? 16 : TX);
  if (strided && TX > 8) TX = 8;               // the X halo grows with the stride: 64-voxel tiles (2 x 4 x 8) keep it in LDS
  int budget = strided ? 64 : 256;
  int rem = budget / TX;
  int TY = p2ceil(g->Y);
  int capy = TX == 16 ? 4 : (strided ? 4 : 8);
  if (TY > capy) TY = capy;
  if (TY > rem) TY = rem;
  int TZ = p2ceil(g->Z);
  if (TZ > rem / TY) TZ = rem / TY;
  g->TZ = TZ, g->TY = TY, g->TX = TX, g->lTX = ilog2(TX), g->lTY = ilog2(TY);
  g->VT = TZ * TY * TX;
  const int HZ = g->sz * (TZ - 1) + 3;
  g->HY = g->sy * (TY - 1) + 3, g->HX = g->sx * (TX - 1) + 3, g->HV = HZ * g->HY * g->HX;
  if (g->VT < 16 || g->VT > RX_WGH_MAX_VT || g->HV > RX_WGH_MAX_HV || HZ > 255 || g->HY > 255 || g->HX > 255) return 0;
  g->tz_n = (g->Z + TZ - 1) / TZ, g->ty_n = (g->Y + TY - 1) / TY, g->tx_n = (g->X + TX - 1) / TX;
  g->NT = g->N * g->tz_n * g->ty_n * g->tx_n;
  g->panels_c = g->Cc / 32;
  const long PP = (long)(g->R / 32) * g->panels_c;
  // the wave-specialised 4x4x16 kernel runs ONE workgroup per CU: 256 workgroups are one full wave of the chip (half the
  // slab traffic and reduce work of 512); the generic kernel runs two per CU
  long target = (!strided && TZ == 4 && TY == 4 && TX == 16 && !getenv("RX_WGH_S512")) ? 256 : 512;
  if (target == 256 && g->NT * PP < 16 * 256) target = 512;   // few tiles (16^3 layers): shorter per-workgroup chains win
  {
    static long tgt_env = -1;   // RX_WGH_TARGET: workgroups aimed at by the split choice of the generic-tile kernel (experiments)
    if (tgt_env < 0) {
      const char* e = getenv("RX_WGH_TARGET");
      tgt_env = e ? atol(e) : 0;
    }
    if (tgt_env > 0 && target == 512) target = tgt_env;
  }
  long S = (target + PP - 1) / PP;
  if (PP >= 256) S = 1;  // the panel pairs alone fill the chip: accumulate every tile in registers, write dw directly
  if (S > g->NT) S = g->NT;
  if (S < 1) S = 1;
  const size_t slab1 = (size_t)27 * g->R * g->Cc * sizeof(float);
  while (S > 1 && S * slab1 > ws_bytes) --S;
  if (S > 1 && slab1 > ws_bytes) return 0;
  g->tiles_per_split = (int)((g->NT + S - 1) / S);
  g->S = (g->NT + g->tiles_per_split - 1) / g->tiles_per_split;
  return 1;
}

size_t rx_wgrad_halo_ws_bytes(const rx_act* x, const rx_act* dy) {
  const long PP = (long)(dy->c / 32) * (x->c / 32);
  long S = PP > 0 ? (512 + PP - 1) / PP : 1;
  return (size_t)S * 27 * dy->c * x->c * sizeof(float) + 256;
}

void rx_wgrad_reduce_launch(const float* slab, int S, int T_, int R, int C, float* dw, hipStream_t st);

// called by rx_conv3d_bwd_weight; returns 1 if it handled the launch, 0 to fall through to the generic kernel,
// negative on error
int rx_wgrad_halo_try(rx_dtype dt, const rx_act* x, const rx_act* dy, const int32_t stride[3], float* dw, void* ws, size_t ws_bytes,
                      hipStream_t st) {
  if (dt == RX_F32) return 0;
  if (x->c % 32 || dy->c % 32 || x->ld % 8 || dy->ld % 8 || ((uintptr_t)x->ptr & 15) || ((uintptr_t)dy->ptr & 15)) return 0;
  WgHaloGeom g;
  if (!wgh_plan(x, dy, stride, &g, ws_bytes)) return 0;
  const size_t lds = (size_t)(RX_WGH_MAX_VT + RX_WGH_MAX_HV) * 64;
  dim3 grid((g.R / 32) * g.panels_c, g.S);
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("RX_DBG");
      dbg = e ? atoi(e) : 0;
    }
    g.dbg = dbg;
    static int order = -1;   // RX_TILE_ORDER: 0 raster, 1 (default) z-fastest walk of each split's tile range (rx_tile_coords)
    if (order < 0) {
      const char* e = getenv("RX_TILE_ORDER");
      order = e ? atoi(e) : 1;
    }
    g.order = order;
  }
  if (g.TZ == 4 && g.TY == 4 && g.TX == 16 && g.sz == 1 && g.sy == 1 && g.sx == 1) {   // compile-time tile
    const size_t lds16 = (size_t)WGH16_BUF_BYTES;
    static bool attr16 = false;
    static int ws_mode = 1, dma_mode = 1;
    if (!attr16) {
      const char* e = getenv("RX_WGH_WS");
      ws_mode = e ? atoi(e) : 1;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16ws_kernel<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds16));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16ws_kernel<f16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds16));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16ws_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds16));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16ws_kernel<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds16));
      const char* ed = getenv("RX_WGH_DMA");
      dma_mode = ed ? atoi(ed) : 1;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo16_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
      attr16 = true;
    }
    rx_note_kernel(ws_mode ? "wgrad_halo16ws_kernel" : "wgrad_halo16_kernel");
    const bool dma_ok = dma_mode && (long)g.N * g.g_ss * 2 < 0x7fffff00L &&
                        ((long)g.N * g.x_ss + (g.x_cs ? (long)(g.Cc / 32 - 1) * g.x_cs : 0)) * 2 < 0x7fffff00L;
    if (ws_mode && dma_ok && dt == RX_BF16)
      hipLaunchKernelGGL((wgrad_halo16ws_kernel<bf16_t, true>), grid, dim3(512), 2 * lds16, st, (const bf16_t*)dy->ptr, (const bf16_t*)x->ptr, (float*)ws, dw, g);
    else if (ws_mode && dma_ok)
      hipLaunchKernelGGL((wgrad_halo16ws_kernel<f16_t, true>), grid, dim3(512), 2 * lds16, st, (const f16_t*)dy->ptr, (const f16_t*)x->ptr, (float*)ws, dw, g);
    else if (ws_mode && dt == RX_BF16)
      hipLaunchKernelGGL((wgrad_halo16ws_kernel<bf16_t, false>), grid, dim3(512), 2 * lds16, st, (const bf16_t*)dy->ptr, (const bf16_t*)x->ptr, (float*)ws, dw, g);
    else if (ws_mode)
      hipLaunchKernelGGL((wgrad_halo16ws_kernel<f16_t, false>), grid, dim3(512), 2 * lds16, st, (const f16_t*)dy->ptr, (const f16_t*)x->ptr, (float*)ws, dw, g);
    else if (dt == RX_BF16)
      hipLaunchKernelGGL((wgrad_halo16_kernel<bf16_t>), grid, dim3(256), lds16, st, (const bf16_t*)dy->ptr, (const bf16_t*)x->ptr, (float*)ws, dw, g);
    else
      hipLaunchKernelGGL((wgrad_halo16_kernel<f16_t>), grid, dim3(256), lds16, st, (const f16_t*)dy->ptr, (const f16_t*)x->ptr, (float*)ws, dw, g);
    if (g.S > 1) rx_wgrad_reduce_launch((const float*)ws, g.S, 27, g.R, g.Cc, dw, st);
    hipError_t e16 = hipGetLastError();
    if (e16 != hipSuccess) {
      rx_set_error("wgrad_halo16: %s", hipGetErrorString(e16));
      return RX_ELAUNCH;
    }
    return 1;
  }
  rx_note_kernel("wgrad_halo_kernel");
  if (dt == RX_BF16) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL((wgrad_halo_kernel<bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)dy->ptr, (const bf16_t*)x->ptr, (float*)ws, dw, g);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL((wgrad_halo_kernel<f16_t>), grid, dim3(256), lds, st, (const f16_t*)dy->ptr, (const f16_t*)x->ptr, (float*)ws, dw, g);
  }
  if (g.S > 1) rx_wgrad_reduce_launch((const float*)ws, g.S, 27, g.R, g.Cc, dw, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rx_set_error("wgrad_halo: %s", hipGetErrorString(e));
    return RX_ELAUNCH;
  }
  return 1;
}
