// rx_loss.hip -- the two task losses of the train step as single-pass HBM-bound kernels.
//
// The torch formulation (reference training/losses/losses.py) makes 5-8 full passes over the (N,C,Z,Y,X) fp32 logits
// per task and direction; here forward = ONE read of logits + target (per-block partial sums -> a one-block fp64
// finalize that also leaves the per-channel coefficients the backward needs on the device), backward = one read of
// both + one write of d(logits).  No host synchronisation: the loss value and the upstream gradient stay device scalars.
//
//   BCEDiceLoss(alpha, beta)   losses.py:307-318
//     bce  = mean over all elements of BCE-with-logits(x, t*(1-2s)+s)             (:217-238, s = 0.1)
//     dice = 1 - mean_c 2*sum(p t) / max(sum(p^2) + sum(t^2), 1e-6),  p = sigmoid(x), sums over (N, spatial)  (:17-43,128-138)
//   MaskedCosineLoss           losses.py:187-215
//     1 - sum(cos(pred/|pred|, t) * m) / (sum(m) + 1e-8),  m = |t| > 1e-6
#include "rx_common.h"

#define RX_LOSS_BLOCK 256
#define RX_LOSS_ELEMS_PER_BLOCK (RX_LOSS_BLOCK * 4 * 8)   // 8 float4 per thread

template <int NACC>
__device__ inline void block_reduce_store(float (&acc)[NACC], float* out) {
  __shared__ float red[NACC][RX_LOSS_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    float v = acc[a];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[a][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RX_LOSS_BLOCK / 64; ++w) s += red[threadIdx.x][w];
    out[threadIdx.x] = s;
  }
}

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// grid (chunks, N*C): plane nc = one (sample, channel) of V contiguous floats
__global__ __launch_bounds__(RX_LOSS_BLOCK) void bce_dice_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, long V,
                                                                          float smoothing, float* __restrict__ partial) {
  const long plane = (long)blockIdx.y * V;
  const long begin = (long)blockIdx.x * RX_LOSS_ELEMS_PER_BLOCK;
  const long end = begin + RX_LOSS_ELEMS_PER_BLOCK < V ? begin + RX_LOSS_ELEMS_PER_BLOCK : V;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // bce, p*t, p*p, t*t
  auto one = [&](float xv, float tv) {
    const float ts = tv * (1.f - 2.f * smoothing) + smoothing;
    const float ax = fabsf(xv);
    acc[0] += fmaxf(xv, 0.f) - xv * ts + log1pf(__expf(-ax));
    const float p = sigmoidf_(xv);
    acc[1] += p * tv;
    acc[2] += p * p;
    acc[3] += tv * tv;
  };
  const bool vec = ((plane & 3) == 0) && ((V & 3) == 0);
  if (vec) {
    for (long i = begin + 4 * threadIdx.x; i < end; i += 4 * RX_LOSS_BLOCK) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + plane + i);
      const f32x4 tv = *reinterpret_cast<const f32x4*>(t + plane + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) one(xv[j], tv[j]);
    }
  } else {
    for (long i = begin + threadIdx.x; i < end; i += RX_LOSS_BLOCK) one(x[plane + i], t[plane + i]);
  }
  block_reduce_store<4>(acc, partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4);
}

// one block: fp64 combination, loss value, per-channel backward coefficients
//   coef[2c]   = a_c = 2 / max(D_c, eps)                 d dice_c / d p = a_c t - b_c p
//   coef[2c+1] = b_c = 4 I_c / D_c^2  (0 where the clamp is active)
__global__ __launch_bounds__(256) void bce_dice_finalize_kernel(const float* __restrict__ partial, int N, int C, int chunks, double count,
                                                                float alpha, float beta, float eps, float* __restrict__ loss,
                                                                float* __restrict__ coef) {
  // one block; per channel the N*chunks partial quadruples are summed by all 256 threads (fp64), then thread 0 combines
  __shared__ double red[4][256];
  double bce = 0.0, dice = 0.0;   // meaningful in thread 0 only
  for (int c = 0; c < C; ++c) {
    double s[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < N * chunks; i += 256) {
      const int n = i / chunks, k = i - n * chunks;
      const float* p = partial + (((long)n * C + c) * chunks + k) * 4;
      s[0] += p[0], s[1] += p[1], s[2] += p[2], s[3] += p[3];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) red[a][threadIdx.x] = s[a];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o)
#pragma unroll
        for (int a = 0; a < 4; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double den = red[2][0] + red[3][0];
      const double D = den > (double)eps ? den : (double)eps;
      bce += red[0][0];
      dice += 2.0 * red[1][0] / D;
      coef[2 * c] = (float)(2.0 / D);
      coef[2 * c + 1] = den > (double)eps ? (float)(4.0 * red[1][0] / (D * D)) : 0.f;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)((double)alpha * (bce / count) + (double)beta * (1.0 - dice / C));
}

__global__ __launch_bounds__(RX_LOSS_BLOCK) void bce_dice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, long V, int C,
                                                                      float smoothing, float k_bce, float k_dice,
                                                                      const float* __restrict__ coef, const float* __restrict__ gloss,
                                                                      float* __restrict__ dx) {
  const int c = blockIdx.y % C;
  const long plane = (long)blockIdx.y * V;
  const long begin = (long)blockIdx.x * RX_LOSS_ELEMS_PER_BLOCK;
  const long end = begin + RX_LOSS_ELEMS_PER_BLOCK < V ? begin + RX_LOSS_ELEMS_PER_BLOCK : V;
  const float g = gloss ? *gloss : 1.f;
  const float a = coef[2 * c], b = coef[2 * c + 1];
  auto one = [&](float xv, float tv) {
    const float ts = tv * (1.f - 2.f * smoothing) + smoothing;
    const float p = sigmoidf_(xv);
    return g * (k_bce * (p - ts) - k_dice * (a * tv - b * p) * p * (1.f - p));
  };
  const bool vec = ((plane & 3) == 0) && ((V & 3) == 0);
  if (vec) {
    for (long i = begin + 4 * threadIdx.x; i < end; i += 4 * RX_LOSS_BLOCK) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + plane + i);
      const f32x4 tv = *reinterpret_cast<const f32x4*>(t + plane + i);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = one(xv[j], tv[j]);
      *reinterpret_cast<f32x4*>(dx + plane + i) = o;
    }
  } else {
    for (long i = begin + threadIdx.x; i < end; i += RX_LOSS_BLOCK) dx[plane + i] = one(x[plane + i], t[plane + i]);
  }
}

// ---- masked cosine: grid (chunks, N); a thread walks voxels, its C channel values sit V apart ---------------------------
#define RX_COS_MAXC 8
__global__ __launch_bounds__(RX_LOSS_BLOCK) void masked_cosine_partial_kernel(const float* __restrict__ pr, const float* __restrict__ tg, long V,
                                                                               int C, float* __restrict__ partial) {
  const long base = (long)blockIdx.y * C * V;
  const long begin = (long)blockIdx.x * RX_LOSS_ELEMS_PER_BLOCK;
  const long end = begin + RX_LOSS_ELEMS_PER_BLOCK < V ? begin + RX_LOSS_ELEMS_PER_BLOCK : V;
  float acc[2] = {0.f, 0.f};  // sum cos*mask, sum mask
  for (long i = begin + threadIdx.x; i < end; i += RX_LOSS_BLOCK) {
    float pp = 0.f, tt = 0.f, pt = 0.f;
    for (int c = 0; c < C; ++c) {
      const float p = pr[base + c * V + i], t = tg[base + c * V + i];
      pp += p * p, tt += t * t, pt += p * t;
    }
    const float pn = sqrtf(pp), tn = sqrtf(tt);
    if (tn > 1e-6f) {
      // u = pred / max(|pred|, 1e-8); cos = (u . t) / (max(|u|, 1e-8) * max(|t|, 1e-8))
      const float pc = fmaxf(pn, 1e-8f);
      const float un = pn / pc;
      acc[0] += (pt / pc) / (fmaxf(un, 1e-8f) * fmaxf(tn, 1e-8f));
      acc[1] += 1.f;
    }
  }
  block_reduce_store<2>(acc, partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2);
}

__global__ __launch_bounds__(256) void masked_cosine_finalize_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ loss,
                                                                     float* __restrict__ coef) {
  __shared__ double s0[256], s1[256];
  double a = 0.0, m = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += partial[2 * i], m += partial[2 * i + 1];
  s0[threadIdx.x] = a, s1[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = m = 0.0;
    for (int i = 0; i < 256; ++i) a += s0[i], m += s1[i];
    *loss = (float)(1.0 - a / (m + 1e-8));
    coef[0] = (float)(1.0 / (m + 1e-8));
  }
}

__global__ __launch_bounds__(RX_LOSS_BLOCK) void masked_cosine_bwd_kernel(const float* __restrict__ pr, const float* __restrict__ tg, long V, int C,
                                                                           const float* __restrict__ coef, const float* __restrict__ gloss,
                                                                           float* __restrict__ dp) {
  const long base = (long)blockIdx.y * C * V;
  const long begin = (long)blockIdx.x * RX_LOSS_ELEMS_PER_BLOCK;
  const long end = begin + RX_LOSS_ELEMS_PER_BLOCK < V ? begin + RX_LOSS_ELEMS_PER_BLOCK : V;
  const float k = -(gloss ? *gloss : 1.f) * coef[0];
  for (long i = begin + threadIdx.x; i < end; i += RX_LOSS_BLOCK) {
    float p[RX_COS_MAXC], t[RX_COS_MAXC];
    float pp = 0.f, tt = 0.f, pt = 0.f;
#pragma unroll
    for (int c = 0; c < RX_COS_MAXC; ++c)
      if (c < C) {
        p[c] = pr[base + c * V + i], t[c] = tg[base + c * V + i];
        pp += p[c] * p[c], tt += t[c] * t[c], pt += p[c] * t[c];
      }
    const float pn = sqrtf(pp), tn = sqrtf(tt);
    // d cos / d pred = (t/|t| - cos * pred/|pred|) / |pred|   (the 1e-8 clamps of the forward are inactive wherever
    // |pred| >= 1e-8; below that the torch graph's gradient is through the clamp constant -- zero direction term)
    const bool on = tn > 1e-6f && pn >= 1e-8f;
    const float inv_p = on ? 1.f / pn : 0.f, inv_t = on ? 1.f / tn : 0.f;
    const float cs = pt * inv_p * inv_t;
#pragma unroll
    for (int c = 0; c < RX_COS_MAXC; ++c)
      if (c < C) dp[base + c * V + i] = on ? k * (t[c] * inv_t - cs * p[c] * inv_p) * inv_p : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static inline int loss_chunks(long V) { return (int)((V + RX_LOSS_ELEMS_PER_BLOCK - 1) / RX_LOSS_ELEMS_PER_BLOCK); }

extern "C" size_t rx_loss_workspace(int n, int c, long v) {
  if (n < 1 || c < 1 || v < 1) return 0;
  return (size_t)n * c * loss_chunks(v) * 4 * sizeof(float) + 256;
}

static int loss_args_ok(const void* a, const void* b, int n, int c, long v) { return a && b && n >= 1 && c >= 1 && v >= 1 && (long)n * c < 65536; }

extern "C" int rx_bce_dice_loss_fwd(const float* logits, const float* target, int n, int c, long v, float alpha, float beta, float smoothing,
                                    float eps, float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
  if (!loss_args_ok(logits, target, n, c, v) || !loss || !coef || !ws) RX_FAIL(RX_EINVAL, "rx_bce_dice_loss_fwd: bad arguments");
  if (ws_bytes < rx_loss_workspace(n, c, v)) RX_FAIL(RX_EWORKSPACE, "rx_bce_dice_loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int chunks = loss_chunks(v);
  hipLaunchKernelGGL(bce_dice_partial_kernel, dim3(chunks, n * c), dim3(RX_LOSS_BLOCK), 0, st, logits, target, v, smoothing, (float*)ws);
  hipLaunchKernelGGL(bce_dice_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, n, c, chunks, (double)n * c * (double)v, alpha,
                     beta, eps, loss, coef);
  RX_CHECK_LAUNCH("rx_bce_dice_loss_fwd");
  return RX_OK;
}

extern "C" int rx_bce_dice_loss_bwd(const float* logits, const float* target, int n, int c, long v, float alpha, float beta, float smoothing,
                                    const float* coef, const float* grad_loss, float* dlogits, void* stream) {
  if (!loss_args_ok(logits, target, n, c, v) || !coef || !dlogits) RX_FAIL(RX_EINVAL, "rx_bce_dice_loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float k_bce = (float)((double)alpha / ((double)n * c * (double)v)), k_dice = beta / c;
  hipLaunchKernelGGL(bce_dice_bwd_kernel, dim3(loss_chunks(v), n * c), dim3(RX_LOSS_BLOCK), 0, st, logits, target, v, c, smoothing, k_bce,
                     k_dice, coef, grad_loss, dlogits);
  RX_CHECK_LAUNCH("rx_bce_dice_loss_bwd");
  return RX_OK;
}

extern "C" int rx_masked_cosine_loss_fwd(const float* pred, const float* target, int n, int c, long v, float* loss, float* coef, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (!loss_args_ok(pred, target, n, c, v) || !loss || !coef || !ws) RX_FAIL(RX_EINVAL, "rx_masked_cosine_loss_fwd: bad arguments");
  if (c > RX_COS_MAXC) RX_FAIL(RX_EUNSUPPORTED, "rx_masked_cosine_loss: at most %d channels (got %d)", RX_COS_MAXC, c);
  if (ws_bytes < rx_loss_workspace(n, c, v)) RX_FAIL(RX_EWORKSPACE, "rx_masked_cosine_loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int chunks = loss_chunks(v);
  hipLaunchKernelGGL(masked_cosine_partial_kernel, dim3(chunks, n), dim3(RX_LOSS_BLOCK), 0, st, pred, target, v, c, (float*)ws);
  hipLaunchKernelGGL(masked_cosine_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, chunks * n, loss, coef);
  RX_CHECK_LAUNCH("rx_masked_cosine_loss_fwd");
  return RX_OK;
}

extern "C" int rx_masked_cosine_loss_bwd(const float* pred, const float* target, int n, int c, long v, const float* coef,
                                         const float* grad_loss, float* dpred, void* stream) {
  if (!loss_args_ok(pred, target, n, c, v) || !coef || !dpred) RX_FAIL(RX_EINVAL, "rx_masked_cosine_loss_bwd: bad arguments");
  if (c > RX_COS_MAXC) RX_FAIL(RX_EUNSUPPORTED, "rx_masked_cosine_loss: at most %d channels (got %d)", RX_COS_MAXC, c);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(masked_cosine_bwd_kernel, dim3(loss_chunks(v), n), dim3(RX_LOSS_BLOCK), 0, st, pred, target, v, c, coef, grad_loss,
                     dpred);
  RX_CHECK_LAUNCH("rx_masked_cosine_loss_bwd");
  return RX_OK;
}
