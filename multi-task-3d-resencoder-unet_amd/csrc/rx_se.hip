// rx_se.hip -- SqueezeExcite + DropPath of the residual blocks (reference call sites: builders/resblocks.py:79-87,
// 109-112 BasicBlockD, :203-212, 234-240 BottleneckD), fused with the InstanceNorm-apply / residual / LeakyReLU passes.
//
// The block computes   a = lrelu( SE( DropPath( xhat ) ) + residual ),   xhat = InstanceNorm(conv_k(..)).
//   DropPath (train only): xhat' = s_n * xhat with the per-sample factor s_n = mask_n / keep_prob (drawn by the caller).
//   SqueezeExcite (dynamic_network_architectures, timm heritage -- the package is NOT in /root/reference: PARITY UNPINNED,
//   restated in oracle/resenc_oracle.py from its published source):
//       p = xhat'.mean((2, 3), keepdim=True)        <- dims 2 and 3 of the input whatever its rank: a 5-D tensor is pooled
//                                                      over (z, y) and KEEPS x (keep_x = 1), a 4-D tensor over (y, x)
//       gate = sigmoid(fc2(relu(fc1(p))))           <- 1x1 convs with bias, rd = make_divisible(C/16, 8) channels
//       out = xhat' * gate
// so a = lrelu(mult * xhat + residual) with the small fp32 tensor mult[n][line][c] = s_n * gate (line = x, or 0).
//
// HBM-bound design: pooling needs only the per-line sums of y (xhat is affine in y given the InstanceNorm statistics), so
//   forward  = ONE extra read of y (line sums: per-block partials, no atomics) + a tiny gate kernel + the usual apply pass;
//   backward = the usual two passes over (g, y, out): line sums of g' and g'*xhat replace the per-(n,c) sums (the
//              InstanceNorm backward means follow from them), a tiny kernel runs the gate backward and the fc gradients.
// With   L1 = sum_zy g',  L2 = sum_zy g'*xhat,  R = voxels per line,  p = raw line mean of xhat:
//   dgate = s*L2;  dz2 = dgate*gate*(1-gate);  dh = W2^T dz2 (relu mask);  dp' = W1^T dh
//   dxhat = g'*mult + D,   D = s*dp'/R;   m1 = mean(dxhat) = sum_lines(mult*L1 + R*D)/V;   m2 = mean(dxhat*xhat) =
//   sum_lines(mult*L2 + D*R*p)/V;   dy = rstd*(dxhat - m1 - xhat*m2);   d_residual (+)= g'.
#include <math.h>

#include "rx_common.h"

template <typename T>
struct SeView {
  const T* ptr;
  long ss;
  int ld;
};
template <typename T>
static inline SeView<T> se_view(const rx_act* a) {
  SeView<T> v;
  v.ptr = (const T*)a->ptr, v.ld = a->ld, v.ss = rx_act_voxels(a) * (long)a->ld;
  return v;
}

// ---- line sums ------------------------------------------------------------------------------------------------------
// grid = (row chunks, x segments, N); thread = one (x, 16-byte channel vector) pair of the segment: every row of the chunk
// is one contiguous, fully coalesced run.  part[((n*chunks + chunk)*NACC + a)*X*C + x*C + c]
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void se_linesum_kernel(SeView<T> y, SeView<T> g, SeView<T> out, const float* __restrict__ stats, int rows, int X,
                                                         int C, int rows_per_chunk, float slope, float* __restrict__ part) {
  constexpr int P = Elem<T>::PER16;
  constexpr int NACC = BWD ? 2 : 1;
  const int CV = C / P;
  const int pair = blockIdx.y * 256 + threadIdx.x;
  if (pair >= X * CV) return;
  const int x = pair / CV, cv = pair - x * CV, n = blockIdx.z;
  float mean[P], rstd[P];
  if (BWD) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      mean[j] = stats[2 * ((size_t)n * C + cv * P + j)];
      rstd[j] = stats[2 * ((size_t)n * C + cv * P + j) + 1];
    }
  }
  float acc[NACC][P];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int j = 0; j < P; ++j) acc[a][j] = 0.f;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
#pragma unroll 4
  for (int r = r0; r < r1; ++r) {
    const long v = (long)r * X + x;
    Vec16<T> yv = ld16(y.ptr + n * y.ss + v * y.ld + cv * P);
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < P; ++j) acc[0][j] += Elem<T>::to_f(yv.v[j]);
    } else {
      Vec16<T> gv = ld16(g.ptr + n * g.ss + v * g.ld + cv * P);
      Vec16<T> ov;
      if (out.ptr) ov = ld16(out.ptr + n * out.ss + v * out.ld + cv * P);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        float gg = Elem<T>::to_f(gv.v[j]);
        const float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
        if (out.ptr && !(Elem<T>::to_f(ov.v[j]) > 0.f)) gg *= slope;
        acc[0][j] += gg;
        acc[NACC - 1][j] += gg * xh;
      }
    }
  }
  float* p = part + ((size_t)(n * gridDim.x + blockIdx.x) * NACC) * X * C + (size_t)x * C + cv * P;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int j = 0; j < P; ++j) p[(size_t)a * X * C + j] = acc[a][j];
}

// sum the partials of all NP planes for every channel of (n, line) into dst[p*C + c] (LDS); all 256 threads take part and
// every load of a thread is issued before the first barrier (the gate kernels are chains of memory round trips)
template <int NP>
__device__ inline void se_gather_line(const float* __restrict__ part, int n, int chunks, int X, int C, int line, int keep_x,
                                      float* __restrict__ dst, float* __restrict__ red) {
  const int tid = threadIdx.x;
  const int XT = keep_x ? 1 : X, nterms = chunks * XT;
  const size_t plane = (size_t)X * C;
  if (C >= 256) {          // one thread per channel (and per further 256 channels): no cross-thread step
    for (int c = tid; c < C; c += 256) {
      float s[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) s[p] = 0.f;
#pragma unroll 8
      for (int t = 0; t < nterms; ++t) {
        const int k = t / XT, x = keep_x ? line : t - k * XT;
        const float* q = part + ((size_t)(n * chunks + k) * NP) * plane + (size_t)x * C + c;
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] += q[p * plane];
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[p * C + c] = s[p];
    }
    __syncthreads();
    return;
  }
  const int KL = 256 / C, cl = tid % C, kl = tid / C;
  float s[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) s[p] = 0.f;
  if (kl < KL) {
#pragma unroll 8
    for (int t = kl; t < nterms; t += KL) {
      const int k = t / XT, x = keep_x ? line : t - k * XT;
      const float* q = part + ((size_t)(n * chunks + k) * NP) * plane + (size_t)x * C + cl;
#pragma unroll
      for (int p = 0; p < NP; ++p) s[p] += q[p * plane];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) red[(p * KL + kl) * C + cl] = s[p];
  }
  __syncthreads();
  for (int i = tid; i < NP * C; i += 256) {
    const int p = i / C, c = i - p * C;
    float tot = 0.f;
    for (int q = 0; q < KL; ++q) tot += red[(p * KL + q) * C + c];
    dst[p * C + c] = tot;
  }
  __syncthreads();
}

// The two small matrix-vector products of a gate kernel.  The deep stages run these kernels with 8-16 workgroups on 512 channels
// and 32 reduction channels: a chain of dependent scalar loads per row was most of their ~23 us.  Both helpers issue every load of
// a thread as independent 16-byte loads before the first use.
//   rows:  out[j] = sum_c W[j*C + c] * v[c],  j < rd      (W row-major over c; v in LDS).  256 / RDP threads share a row.
template <int RDP>
__device__ inline void se_mv_rows(const float* __restrict__ W, const float* __restrict__ v, int C, int rd, float* __restrict__ out) {
  constexpr int TPR = 256 / RDP;                 // threads per row (4 .. 32): consecutive lanes of one wave
  const int tid = threadIdx.x, j = tid / TPR, part = tid - j * TPR;
  float a = 0.f;
  if (j < rd) {
    const float* wr = W + (size_t)j * C;
    for (int c0 = part * 4; c0 < C; c0 += TPR * 4 * 4) {
      f32x4 w4[4], v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * TPR * 4;
        w4[u] = c < C ? *reinterpret_cast<const f32x4*>(wr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        v4[u] = c < C ? *reinterpret_cast<const f32x4*>(v + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) a += w4[u][0] * v4[u][0] + w4[u][1] * v4[u][1] + w4[u][2] * v4[u][2] + w4[u][3] * v4[u][3];
    }
  }
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) a += __shfl_xor(a, o, 64);
  if (part == 0 && j < rd) out[j] = a;
}
__device__ inline void se_mv_rows_any(const float* __restrict__ W, const float* __restrict__ v, int C, int rd, float* __restrict__ out) {
  if (rd <= 8) se_mv_rows<8>(W, v, C, rd, out);
  else if (rd <= 16) se_mv_rows<16>(W, v, C, rd, out);
  else if (rd <= 32) se_mv_rows<32>(W, v, C, rd, out);
  else se_mv_rows<64>(W, v, C, rd, out);
}
//   per channel:  sum_j W[c*rd + j] * h[j]  for one channel c (W row-major over j, rd % 4 == 0; h in LDS)
__device__ inline float se_dot_row(const float* __restrict__ wr, const float* __restrict__ h, int rd) {
  float z = 0.f;
  for (int j0 = 0; j0 < rd; j0 += 16) {
    f32x4 w4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w4[u] = j0 + 4 * u < rd ? *reinterpret_cast<const f32x4*>(wr + j0 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + 4 * u < rd) {
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(h + j0 + 4 * u);
        z += w4[u][0] * h4[0] + w4[u][1] * h4[1] + w4[u][2] * h4[2] + w4[u][3] * h4[3];
      }
  }
  return z;
}

// ---- gate forward: grid = (lines, N) ----------------------------------------------------------------------------------
// LDS: sp[C] | sh[64] | red[256]
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* __restrict__ part, int chunks, int X, int C, int keep_x, float R,
                                                          const float* __restrict__ stats, const float* __restrict__ path_scale,
                                                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int rd, float* __restrict__ pooled, float* __restrict__ hidden,
                                                          float* __restrict__ gate, float* __restrict__ mult) {
  extern __shared__ float sm[];
  float* sp = sm;
  float* sh = sm + C;
  float* red = sh + 64;
  const int line = blockIdx.x, n = blockIdx.y, L = gridDim.x, tid = threadIdx.x;
  const float s = path_scale ? path_scale[n] : 1.f;
  const size_t row = (size_t)n * L + line;
  // 16-byte loads of the fc weights: channel counts / reduction widths that are multiples of 4, 16-byte aligned pointers
  const bool vec_ok = (C % 4 == 0) && (rd % 4 == 0) && !(((uintptr_t)w1 | (uintptr_t)w2) & 15);
  se_gather_line<1>(part, n, chunks, X, C, line, keep_x, sp, red);
  for (int c = tid; c < C; c += 256) {
    const float praw = (sp[c] / R - stats[2 * ((size_t)n * C + c)]) * stats[2 * ((size_t)n * C + c) + 1];
    pooled[row * C + c] = praw;
    sp[c] = s * praw;
  }
  __syncthreads();
  if (vec_ok) {
    se_mv_rows_any(w1, sp, C, rd, sh);
    __syncthreads();
    if (tid < rd) {
      float a = sh[tid] + b1[tid];
      a = a > 0.f ? a : 0.f;
      sh[tid] = a;
      hidden[row * rd + tid] = a;
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < rd; j += 4) {
      float a = 0.f;
      for (int c = lane; c < C; c += 64) a += w1[(size_t)j * C + c] * sp[c];
      a = wave_sum(a);
      if (lane == 0) {
        a += b1[j];
        a = a > 0.f ? a : 0.f;
        sh[j] = a;
        hidden[row * rd + j] = a;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float z = b2[c];
    if (vec_ok)
      z += se_dot_row(w2 + (size_t)c * rd, sh, rd);
    else
      for (int j = 0; j < rd; ++j) z += w2[(size_t)c * rd + j] * sh[j];
    const float gt = 1.f / (1.f + expf(-z));
    gate[row * C + c] = gt;
    mult[row * C + c] = s * gt;
  }
}

// DropPath without SE: mult = s_n
__global__ __launch_bounds__(256) void se_fill_mult_kernel(const float* __restrict__ path_scale, int per_sample, long total, float* __restrict__ mult) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < total) mult[i] = path_scale ? path_scale[i / per_sample] : 1.f;
}

// ---- gate backward: grid = (lines, N) ---------------------------------------------------------------------------------
// LDS: sL1[C] | sL2[C] | sds[C] | sdh[64] | red[512]
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* __restrict__ part, int chunks, int X, int C, int keep_x, float R,
                                                          const float* __restrict__ path_scale, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int rd, const float* __restrict__ pooled,
                                                          const float* __restrict__ hidden, const float* __restrict__ gate,
                                                          const float* __restrict__ mult, float* __restrict__ dadd, float* __restrict__ dz2,
                                                          float* __restrict__ dhm, float* __restrict__ line_m) {
  extern __shared__ float sm[];
  float* sL1 = sm;
  float* sL2 = sm + C;
  float* sds = sm + 2 * C;
  float* sdh = sm + 3 * C;
  float* red = sdh + 64;
  const int line = blockIdx.x, n = blockIdx.y, L = gridDim.x, tid = threadIdx.x;
  const float s = path_scale ? path_scale[n] : 1.f;
  const size_t row = (size_t)n * L + line;
  se_gather_line<2>(part, n, chunks, X, C, line, keep_x, sL1, red);     // sL2 = sL1 + C
  if (w1 == nullptr) {   // DropPath only
    for (int c = tid; c < C; c += 256) {
      const float m = mult[row * C + c];
      dadd[row * C + c] = 0.f;
      line_m[(row * 2 + 0) * C + c] = m * sL1[c];
      line_m[(row * 2 + 1) * C + c] = m * sL2[c];
    }
    return;
  }
  for (int c = tid; c < C; c += 256) {
    const float gt = gate[row * C + c];
    const float d = s * sL2[c] * gt * (1.f - gt);
    sds[c] = d;
    dz2[row * C + c] = d;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  if ((rd % 4 == 0) && rd <= 64 && !((uintptr_t)w2 & 15)) {
    // dh[j] = sum_c W2[c][j] * ds[c]: a thread takes whole ROWS of W2 (16-byte loads, rd contiguous floats per channel) and keeps
    // rd partial sums; one xor-shuffle tree per j and a cross-wave add finish.  (A wave per j read W2 with a stride of rd floats:
    // 64 separate cache lines per load instruction.)
    float pj[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) pj[j] = 0.f;
    for (int c = tid; c < C; c += 256) {
      const float d = sds[c];
      const float* wr = w2 + (size_t)c * rd;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (4 * q < rd) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * q);
          pj[4 * q] += w4[0] * d, pj[4 * q + 1] += w4[1] * d, pj[4 * q + 2] += w4[2] * d, pj[4 * q + 3] += w4[3] * d;
        }
    }
#pragma unroll
    for (int j = 0; j < 64; ++j)
      if (j < rd) {
        float a = wave_sum(pj[j]);
        if (lane == 0) red[wave * 64 + j] = a;
      }
    __syncthreads();
    if (tid < rd) {
      float a = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
      a = hidden[row * rd + tid] > 0.f ? a : 0.f;
      sdh[tid] = a;
      dhm[row * rd + tid] = a;
    }
  } else {
    for (int j = wave; j < rd; j += 4) {
      float a = 0.f;
      for (int c = lane; c < C; c += 64) a += w2[(size_t)c * rd + j] * sds[c];
      a = wave_sum(a);
      if (lane == 0) {
        a = hidden[row * rd + j] > 0.f ? a : 0.f;
        sdh[j] = a;
        dhm[row * rd + j] = a;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float dp = 0.f;
    for (int j0 = 0; j0 < rd; j0 += 32) {         // up to 32 coalesced loads in flight (adjacent threads = adjacent channels)
      float wv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) wv[u] = j0 + u < rd ? w1[(size_t)(j0 + u) * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 32; ++u)
        if (j0 + u < rd) dp += wv[u] * sdh[j0 + u];
    }
    const float D = s * dp / R, m = mult[row * C + c];
    dadd[row * C + c] = D;
    line_m[(row * 2 + 0) * C + c] = m * sL1[c] + R * D;
    line_m[(row * 2 + 1) * C + c] = m * sL2[c] + D * R * pooled[row * C + c];
  }
}

// ---- gate backward, deep-stage shape (C <= 512, rd <= 32, rd % 4 == 0): every global value a thread needs -- its two channels' gate /
// mult / pooled, its W2 rows and W1 columns, the hidden row -- is requested BEFORE the partial sums are gathered, so the six phases
// below only wait for LDS and shuffles.  Same formulas and summation order as se_gate_bwd_kernel: dadd, dz2, dhm come out bit-identical,
// line_m within one ulp (the compiler fuses a different product of m * L + R * D); scripts/probes/se_bwd_ab.py compares two builds.
__global__ __launch_bounds__(256) void se_gate_bwd_small_kernel(const float* __restrict__ part, int chunks, int X, int C, int keep_x, float R,
                                                                const float* __restrict__ path_scale, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, int rd, const float* __restrict__ pooled,
                                                                const float* __restrict__ hidden, const float* __restrict__ gate,
                                                                const float* __restrict__ mult, float* __restrict__ dadd,
                                                                float* __restrict__ dz2, float* __restrict__ dhm, float* __restrict__ line_m) {
  extern __shared__ float sm[];
  float* sL1 = sm;
  float* sL2 = sm + C;
  float* sds = sm + 2 * C;
  float* sdh = sm + 3 * C;
  float* red = sdh + 64;
  const int line = blockIdx.x, n = blockIdx.y, L = gridDim.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float s = path_scale ? path_scale[n] : 1.f;
  const size_t row = (size_t)n * L + line;
  // ---- prefetch
  float gt_r[2], m_r[2], p_r[2], w1c[2][32];
  f32x4 w2r[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    const bool on = c < C;
    gt_r[k] = on ? gate[row * C + c] : 0.f;
    m_r[k] = on ? mult[row * C + c] : 0.f;
    p_r[k] = on ? pooled[row * C + c] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) w2r[k][q] = (on && 4 * q < rd) ? *reinterpret_cast<const f32x4*>(w2 + (size_t)c * rd + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 32; ++j) w1c[k][j] = (on && j < rd) ? w1[(size_t)j * C + c] : 0.f;
  }
  const float hid = tid < rd ? hidden[row * rd + tid] : 0.f;
  se_gather_line<2>(part, n, chunks, X, C, line, keep_x, sL1, red);     // sL2 = sL1 + C
  float ds_r[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    ds_r[k] = 0.f;
    if (c < C) {
      ds_r[k] = s * sL2[c] * gt_r[k] * (1.f - gt_r[k]);
      dz2[row * C + c] = ds_r[k];
    }
  }
  float pj[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) pj[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      pj[4 * q] += w2r[k][q][0] * ds_r[k], pj[4 * q + 1] += w2r[k][q][1] * ds_r[k];
      pj[4 * q + 2] += w2r[k][q][2] * ds_r[k], pj[4 * q + 3] += w2r[k][q][3] * ds_r[k];
    }
#pragma unroll
  for (int j = 0; j < 32; ++j)
    if (j < rd) {
      const float a = wave_sum(pj[j]);
      if (lane == 0) red[wave * 64 + j] = a;
    }
  __syncthreads();
  if (tid < rd) {
    float a = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
    a = hid > 0.f ? a : 0.f;
    sdh[tid] = a;
    dhm[row * rd + tid] = a;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    if (c < C) {
      float dp = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j)
        if (j < rd) dp += w1c[k][j] * sdh[j];
      const float D = s * dp / R;
      dadd[row * C + c] = D;
      line_m[(row * 2 + 0) * C + c] = m_r[k] * sL1[c] + R * D;
      line_m[(row * 2 + 1) * C + c] = m_r[k] * sL2[c] + D * R * p_r[k];
    }
  }
}

// m12[n][c] = (sum_lines t1, sum_lines t2) / V
__global__ __launch_bounds__(256) void se_m12_kernel(const float* __restrict__ line_m, int N, int L, int C, double V, float* __restrict__ m12) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double a = 0.0, b = 0.0;
  int l = 0;
  for (; l + 8 <= L; l += 8) {          // 16 loads in flight, added in line order
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      va[u] = line_m[(((size_t)n * L + l + u) * 2 + 0) * C + c];
      vb[u] = line_m[(((size_t)n * L + l + u) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (double)va[u], b += (double)vb[u];
  }
  for (; l < L; ++l) {
    a += (double)line_m[(((size_t)n * L + l) * 2 + 0) * C + c];
    b += (double)line_m[(((size_t)n * L + l) * 2 + 1) * C + c];
  }
  m12[2 * i] = (float)(a / V);
  m12[2 * i + 1] = (float)(b / V);
}

// fc gradients: one wave per output element, lanes stride over the (n, line) rows; fixed order, fp64 combine
// outputs: dw1 (rd*C) | db1 (rd) | dw2 (C*rd) | db2 (C)
__global__ __launch_bounds__(256) void se_param_grad_kernel(const float* __restrict__ dz2, const float* __restrict__ dhm, const float* __restrict__ pooled,
                                                            const float* __restrict__ hidden, const float* __restrict__ path_scale, int N, int L,
                                                            int C, int rd, float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n1 = rd * C, n2 = n1 + rd, n3 = n2 + C * rd, n4 = n3 + C;
  if (i >= n4) return;
  double a = 0.0;
  const int rows = N * L;
  if (i < n1) {
    const int j = i / C, c = i - j * C;
    for (int r = lane; r < rows; r += 64) a += (double)dhm[(size_t)r * rd + j] * (double)((path_scale ? path_scale[r / L] : 1.f) * pooled[(size_t)r * C + c]);
  } else if (i < n2) {
    const int j = i - n1;
    for (int r = lane; r < rows; r += 64) a += (double)dhm[(size_t)r * rd + j];
  } else if (i < n3) {
    const int k = i - n2, c = k / rd, j = k - c * rd;
    for (int r = lane; r < rows; r += 64) a += (double)dz2[(size_t)r * C + c] * (double)hidden[(size_t)r * rd + j];
  } else {
    const int c = i - n3;
    for (int r = lane; r < rows; r += 64) a += (double)dz2[(size_t)r * C + c];
  }
  a = wave_sum_d(a);
  if (lane != 0) return;
  if (i < n1) dw1[i] = (float)a;
  else if (i < n2) db1[i - n1] = (float)a;
  else if (i < n3) dw2[i - n2] = (float)a;
  else db2[i - n3] = (float)a;
}

// ---- gated apply passes ---------------------------------------------------------------------------------------------
template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void in_gate_act_fwd_kernel(SeView<T> y, const float* __restrict__ stats, const float* __restrict__ mult, int L, int X,
                                                              SeView<T> res, T* __restrict__ out, int ldo, long so, int V, int C, float slope) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P, n = blockIdx.y;
  const long total = (long)V * CV, step = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    mean[j] = stats[2 * ((size_t)n * C + cv * P + j)];
    rstd[j] = stats[2 * ((size_t)n * C + cv * P + j) + 1];
  }
  for (; i < total; i += step) {
    const long v = i / CV;
    const int line = L == 1 ? 0 : (int)(v % X);
    const float* mp = mult + ((size_t)n * L + line) * C + cv * P;
    Vec16<T> a = ld16(y.ptr + n * y.ss + v * y.ld + cv * P), r, o;
    if (HAS_RES) r = ld16(res.ptr + n * res.ss + v * res.ld + cv * P);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float f = (Elem<T>::to_f(a.v[j]) - mean[j]) * rstd[j] * mp[j];
      if (HAS_RES) f += Elem<T>::to_f(r.v[j]);
      f = f > 0.f ? f : f * slope;
      o.v[j] = Elem<T>::from_f(f);
    }
    st16(out + n * so + v * ldo + cv * P, o);
  }
}

template <typename T, bool HAS_DRES, bool ACC_DRES>
__global__ __launch_bounds__(256) void in_gate_act_bwd_kernel(SeView<T> g, SeView<T> y, SeView<T> out, const float* __restrict__ stats,
                                                              const float* __restrict__ mult, const float* __restrict__ dadd,
                                                              const float* __restrict__ m12, int L, int X, T* __restrict__ dy, int lddy, long sdy,
                                                              T* __restrict__ dres, int lddr, long sdr, int V, int C, float slope) {
  constexpr int P = Elem<T>::PER16;
  const int CV = C / P, n = blockIdx.y;
  const long total = (long)V * CV, step = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int cv = (int)(i % CV);
  float mean[P], rstd[P], m1[P], m2[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const size_t k = (size_t)n * C + cv * P + j;
    mean[j] = stats[2 * k], rstd[j] = stats[2 * k + 1];
    m1[j] = m12[2 * k], m2[j] = m12[2 * k + 1];
  }
  for (; i < total; i += step) {
    const long v = i / CV;
    const int line = L == 1 ? 0 : (int)(v % X);
    const float* mp = mult + ((size_t)n * L + line) * C + cv * P;
    const float* dp = dadd + ((size_t)n * L + line) * C + cv * P;
    Vec16<T> gv = ld16(g.ptr + n * g.ss + v * g.ld + cv * P);
    Vec16<T> yv = ld16(y.ptr + n * y.ss + v * y.ld + cv * P);
    Vec16<T> ov, dv, rv;
    if (out.ptr) ov = ld16(out.ptr + n * out.ss + v * out.ld + cv * P);
    if (HAS_DRES && ACC_DRES) rv = ld16(dres + n * sdr + v * lddr + cv * P);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float gg = Elem<T>::to_f(gv.v[j]);
      const float xh = (Elem<T>::to_f(yv.v[j]) - mean[j]) * rstd[j];
      if (out.ptr && !(Elem<T>::to_f(ov.v[j]) > 0.f)) gg *= slope;
      const float dx = gg * mp[j] + dp[j];
      dv.v[j] = Elem<T>::from_f(rstd[j] * (dx - m1[j] - xh * m2[j]));
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

// ---- host -----------------------------------------------------------------------------------------------------------
struct SePlan {
  int rows, X, L, chunks, rows_per_chunk, segs;
  float R;
};
static SePlan se_plan(const rx_act* y, int per16, int keep_x) {
  SePlan p;
  p.rows = y->z * y->y, p.X = y->x;
  p.L = keep_x ? y->x : 1;
  p.R = keep_x ? (float)p.rows : (float)p.rows * (float)y->x;
  const int CV = y->c / per16;
  p.segs = (p.X * CV + 255) / 256;
  // a line-sum thread should stream >= 8 rows, and the gate kernels add `chunks` partials per channel one after the other:
  // at most 128 chunks (128^3 x 32 channels, batch 2: 512 blocks of 128 rows)
  long want = p.rows / 8;
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  p.rows_per_chunk = (int)((p.rows + want - 1) / want);
  p.chunks = (p.rows + p.rows_per_chunk - 1) / p.rows_per_chunk;
  return p;
}
static int se_check(const rx_act* a, int dt, const char* who) {
  const int per16 = dt == RX_F32 ? 4 : 8;
  if (!rx_act_ok(a)) RX_FAIL(RX_EINVAL, "%s: bad activation descriptor", who);
  if (a->c % per16 || a->ld % per16 || ((uintptr_t)a->ptr & 15)) RX_FAIL(RX_EUNSUPPORTED, "%s: channels/ld/ptr must be 16-byte multiples (c=%d ld=%d)", who, a->c, a->ld);
  if (a->c > 2048) RX_FAIL(RX_EUNSUPPORTED, "%s: too many channels (%d)", who, a->c);
  return RX_OK;
}
static int se_same(const rx_act* a, const rx_act* b) { return a->n == b->n && a->z == b->z && a->y == b->y && a->x == b->x && a->c == b->c; }
static inline size_t se_al(size_t v) { return (v + 63) & ~(size_t)63; }

// workspace: line-sum partials (2 planes) | dz2 [N][L][C] | dhm [N][L][64] | line_m [N][L][2][C]
extern "C" size_t rx_se_workspace(const rx_act* y) {
  if (!rx_act_ok(y)) return 0;
  const size_t N = y->n, X = y->x, C = y->c;
  return se_al(N * 256 * 2 * X * C * sizeof(float)) + se_al(N * X * C * sizeof(float)) + se_al(N * X * 64 * sizeof(float)) +
         se_al(N * X * 2 * C * sizeof(float)) + 256;
}

extern "C" int rx_se_gate_fwd(rx_dtype dt, const rx_act* y, const float* stats, const float* path_scale, const rx_se_params* se, float* pooled,
                              float* hidden, float* gate, float* mult, void* ws, size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), se_ = RxSeV(se)](void* s) { return rx_se_gate_fwd(dt, y_.p(), stats, path_scale, se_.p(), pooled, hidden, gate, mult, ws, ws_bytes, s); });
  int rc = se_check(y, dt, "rx_se_gate_fwd(y)");
  if (rc) return rc;
  if (!mult) RX_FAIL(RX_EINVAL, "rx_se_gate_fwd: null mult");
  hipStream_t st = (hipStream_t)stream;
  if (!se) {   // DropPath only
    const int L = y->x;   // mult is laid out [n][x][c] like the SE case (keep_x = 1)
    const long total = (long)y->n * L * y->c;
    hipLaunchKernelGGL(se_fill_mult_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, path_scale, L * y->c, total, mult);
    RX_CHECK_LAUNCH("rx_se_gate_fwd(fill)");
    return RX_OK;
  }
  if (!stats || !pooled || !hidden || !gate || !ws || !se->w1 || !se->b1 || !se->w2 || !se->b2) RX_FAIL(RX_EINVAL, "rx_se_gate_fwd: null argument");
  if (se->rd < 1 || se->rd > 64) RX_FAIL(RX_EUNSUPPORTED, "rx_se_gate_fwd: reduction channels %d outside [1, 64]", se->rd);
  if (ws_bytes < rx_se_workspace(y)) RX_FAIL(RX_EWORKSPACE, "rx_se_gate_fwd: workspace too small");
  const int per16 = dt == RX_F32 ? 4 : 8;
  const SePlan p = se_plan(y, per16, se->keep_x);
  float* part = (float*)ws;
  RX_DISPATCH_DTYPE(dt, T, {
    SeView<T> yv = se_view<T>(y), none{nullptr, 0, 0};
    hipLaunchKernelGGL((se_linesum_kernel<T, false>), dim3(p.chunks, p.segs, y->n), dim3(256), 0, st, yv, none, none, stats, p.rows, p.X, y->c,
                       p.rows_per_chunk, 1.f, part);
  });
  const size_t lds = (size_t)(y->c + 64 + 256) * sizeof(float);
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(p.L, y->n), dim3(256), lds, st, (const float*)part, p.chunks, p.X, y->c, se->keep_x, p.R, stats,
                     path_scale, se->w1, se->b1, se->w2, se->b2, se->rd, pooled, hidden, gate, mult);
  RX_CHECK_LAUNCH("rx_se_gate_fwd");
  return RX_OK;
}

extern "C" int rx_instnorm_gate_act_fwd(rx_dtype dt, const rx_act* y, const float* stats, const float* mult, int keep_x, const rx_act* residual,
                                        const rx_act* out, float slope, void* stream) {
  RX_RECORD(stream, [=, y_ = RxActV(y), residual_ = RxActV(residual), out_ = RxActV(out)](void* s) { return rx_instnorm_gate_act_fwd(dt, y_.p(), stats, mult, keep_x, residual_.p(), out_.p(), slope, s); });
  int rc;
  if ((rc = se_check(y, dt, "rx_instnorm_gate_act_fwd(y)"))) return rc;
  if ((rc = se_check(out, dt, "rx_instnorm_gate_act_fwd(out)"))) return rc;
  if (!stats || !mult || !se_same(y, out)) RX_FAIL(RX_EINVAL, "rx_instnorm_gate_act_fwd: bad arguments");
  if (residual) {
    if ((rc = se_check(residual, dt, "rx_instnorm_gate_act_fwd(residual)"))) return rc;
    if (!se_same(y, residual)) RX_FAIL(RX_EINVAL, "rx_instnorm_gate_act_fwd: residual geometry mismatch");
  }
  const long V = rx_act_voxels(y);
  hipStream_t st = (hipStream_t)stream;
  const int L = keep_x ? y->x : 1;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    const int CV = y->c / P;
    long want = (V * CV + 2047) / 2048;
    int gq = CV;                                                 // G * 256 % CV == 0: a thread keeps one channel vector
    for (int d = 256; gq % 2 == 0 && d > 1; d /= 2) gq /= 2;
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;
    const int G = (int)((want + gq - 1) / gq * gq);
    SeView<T> yv = se_view<T>(y);
    if (residual)
      hipLaunchKernelGGL((in_gate_act_fwd_kernel<T, true>), dim3(G, y->n), dim3(256), 0, st, yv, stats, mult, L, y->x, se_view<T>(residual),
                         (T*)out->ptr, out->ld, V * out->ld, (int)V, y->c, slope);
    else
      hipLaunchKernelGGL((in_gate_act_fwd_kernel<T, false>), dim3(G, y->n), dim3(256), 0, st, yv, stats, mult, L, y->x, SeView<T>{nullptr, 0, 0},
                         (T*)out->ptr, out->ld, V * out->ld, (int)V, y->c, slope);
  });
  RX_CHECK_LAUNCH("rx_instnorm_gate_act_fwd");
  return RX_OK;
}

extern "C" int rx_se_gate_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out, float slope,
                              const float* path_scale, const rx_se_params* se, const float* pooled, const float* hidden, const float* gate,
                              const float* mult, float* dadd, float* m12, float* dw1, float* db1, float* dw2, float* db2, void* ws,
                              size_t ws_bytes, void* stream) {
  RX_RECORD(stream, [=, g_ = RxActV(g), y_ = RxActV(y), out_ = RxActV(out), se_ = RxSeV(se)](void* s) { return rx_se_gate_bwd(dt, g_.p(), y_.p(), stats, out_.p(), slope, path_scale, se_.p(), pooled, hidden, gate, mult, dadd, m12, dw1, db1, dw2, db2, ws, ws_bytes, s); });
  int rc;
  if ((rc = se_check(g, dt, "rx_se_gate_bwd(g)"))) return rc;
  if ((rc = se_check(y, dt, "rx_se_gate_bwd(y)"))) return rc;
  if (!se_same(g, y) || !stats || !mult || !dadd || !m12 || !ws) RX_FAIL(RX_EINVAL, "rx_se_gate_bwd: bad arguments");
  if (out) {
    if ((rc = se_check(out, dt, "rx_se_gate_bwd(out)"))) return rc;
    if (!se_same(out, y)) RX_FAIL(RX_EINVAL, "rx_se_gate_bwd: out geometry mismatch");
  }
  if (se && (!pooled || !hidden || !gate || !dw1 || !db1 || !dw2 || !db2 || !se->w1 || !se->w2)) RX_FAIL(RX_EINVAL, "rx_se_gate_bwd: null SE argument");
  if (se && (se->rd < 1 || se->rd > 64)) RX_FAIL(RX_EUNSUPPORTED, "rx_se_gate_bwd: reduction channels %d outside [1, 64]", se->rd);
  if (ws_bytes < rx_se_workspace(y)) RX_FAIL(RX_EWORKSPACE, "rx_se_gate_bwd: workspace too small");
  const int per16 = dt == RX_F32 ? 4 : 8;
  const int keep_x = se ? se->keep_x : 1;
  const SePlan p = se_plan(y, per16, keep_x);
  const size_t N = y->n, X = y->x, C = y->c;
  char* w = (char*)ws;
  float* part = (float*)w;
  w += se_al(N * 256 * 2 * X * C * sizeof(float));
  float* dz2 = (float*)w;
  w += se_al(N * X * C * sizeof(float));
  float* dhm = (float*)w;
  w += se_al(N * X * 64 * sizeof(float));
  float* line_m = (float*)w;
  hipStream_t st = (hipStream_t)stream;
  const bool masked = out && slope != 1.f;
  RX_DISPATCH_DTYPE(dt, T, {
    SeView<T> ov = masked ? se_view<T>(out) : SeView<T>{nullptr, 0, 0};
    hipLaunchKernelGGL((se_linesum_kernel<T, true>), dim3(p.chunks, p.segs, y->n), dim3(256), 0, st, se_view<T>(y), se_view<T>(g), ov, stats, p.rows,
                       p.X, y->c, p.rows_per_chunk, slope, part);
  });
  const size_t lds = (size_t)(3 * C + 64 + 512) * sizeof(float);
  if (se && se->w1 && y->c <= 512 && se->rd <= 32 && se->rd % 4 == 0 && !((uintptr_t)se->w2 & 15))
    hipLaunchKernelGGL(se_gate_bwd_small_kernel, dim3(p.L, y->n), dim3(256), lds, st, (const float*)part, p.chunks, p.X, y->c, keep_x, p.R, path_scale,
                     se ? se->w1 : (const float*)nullptr, se ? se->w2 : (const float*)nullptr, se ? se->rd : 0, pooled, hidden, gate, mult, dadd, dz2,
                     dhm, line_m);
  else
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(p.L, y->n), dim3(256), lds, st, (const float*)part, p.chunks, p.X, y->c, keep_x, p.R, path_scale,
                     se ? se->w1 : (const float*)nullptr, se ? se->w2 : (const float*)nullptr, se ? se->rd : 0, pooled, hidden, gate, mult, dadd, dz2,
                     dhm, line_m);
  hipLaunchKernelGGL(se_m12_kernel, dim3((y->n * y->c + 255) / 256), dim3(256), 0, st, (const float*)line_m, y->n, p.L, y->c,
                     (double)rx_act_voxels(y), m12);
  if (se) {
    const int nout = 2 * se->rd * y->c + se->rd + y->c;
    hipLaunchKernelGGL(se_param_grad_kernel, dim3((nout + 3) / 4), dim3(256), 0, st, (const float*)dz2, (const float*)dhm, pooled, hidden, path_scale,
                       y->n, p.L, y->c, se->rd, dw1, db1, dw2, db2);
  }
  RX_CHECK_LAUNCH("rx_se_gate_bwd");
  return RX_OK;
}

extern "C" int rx_instnorm_gate_act_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out, float slope,
                                        const float* mult, const float* dadd, const float* m12, int keep_x, const rx_act* dy,
                                        const rx_act* d_residual, int accumulate_residual, void* stream) {
  RX_RECORD(stream, [=, g_ = RxActV(g), y_ = RxActV(y), out_ = RxActV(out), dy_ = RxActV(dy), d_residual_ = RxActV(d_residual)](void* s) { return rx_instnorm_gate_act_bwd(dt, g_.p(), y_.p(), stats, out_.p(), slope, mult, dadd, m12, keep_x, dy_.p(), d_residual_.p(), accumulate_residual, s); });
  int rc;
  if ((rc = se_check(g, dt, "rx_instnorm_gate_act_bwd(g)"))) return rc;
  if ((rc = se_check(y, dt, "rx_instnorm_gate_act_bwd(y)"))) return rc;
  if ((rc = se_check(dy, dt, "rx_instnorm_gate_act_bwd(dy)"))) return rc;
  if (!se_same(g, y) || !se_same(dy, y) || !stats || !mult || !dadd || !m12) RX_FAIL(RX_EINVAL, "rx_instnorm_gate_act_bwd: bad arguments");
  if (out) {
    if ((rc = se_check(out, dt, "rx_instnorm_gate_act_bwd(out)"))) return rc;
    if (!se_same(out, y)) RX_FAIL(RX_EINVAL, "rx_instnorm_gate_act_bwd: out geometry mismatch");
  }
  if (d_residual) {
    if ((rc = se_check(d_residual, dt, "rx_instnorm_gate_act_bwd(d_residual)"))) return rc;
    if (!se_same(d_residual, y)) RX_FAIL(RX_EINVAL, "rx_instnorm_gate_act_bwd: d_residual geometry mismatch");
  }
  const long V = rx_act_voxels(y);
  hipStream_t st = (hipStream_t)stream;
  const int L = keep_x ? y->x : 1;
  const bool masked = out && slope != 1.f;
  RX_DISPATCH_DTYPE(dt, T, {
    constexpr int P = Elem<T>::PER16;
    const int CV = y->c / P;
    long want = (V * CV + 2047) / 2048;
    int gq = CV;
    for (int d = 256; gq % 2 == 0 && d > 1; d /= 2) gq /= 2;
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;
    const int G = (int)((want + gq - 1) / gq * gq);
    SeView<T> ov = masked ? se_view<T>(out) : SeView<T>{nullptr, 0, 0};
    T* dr = d_residual ? (T*)d_residual->ptr : (T*)nullptr;
    const int lddr = d_residual ? d_residual->ld : 0;
    const long sdr = d_residual ? V * d_residual->ld : 0L;
    if (!d_residual)
      hipLaunchKernelGGL((in_gate_act_bwd_kernel<T, false, false>), dim3(G, y->n), dim3(256), 0, st, se_view<T>(g), se_view<T>(y), ov, stats, mult, dadd,
                         m12, L, y->x, (T*)dy->ptr, dy->ld, V * dy->ld, dr, lddr, sdr, (int)V, y->c, slope);
    else if (accumulate_residual)
      hipLaunchKernelGGL((in_gate_act_bwd_kernel<T, true, true>), dim3(G, y->n), dim3(256), 0, st, se_view<T>(g), se_view<T>(y), ov, stats, mult, dadd,
                         m12, L, y->x, (T*)dy->ptr, dy->ld, V * dy->ld, dr, lddr, sdr, (int)V, y->c, slope);
    else
      hipLaunchKernelGGL((in_gate_act_bwd_kernel<T, true, false>), dim3(G, y->n), dim3(256), 0, st, se_view<T>(g), se_view<T>(y), ov, stats, mult, dadd,
                         m12, L, y->x, (T*)dy->ptr, dy->ld, V * dy->ld, dr, lddr, sdr, (int)V, y->c, slope);
  });
  RX_CHECK_LAUNCH("rx_instnorm_gate_act_bwd");
  return RX_OK;
}
