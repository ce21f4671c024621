/* rxunet.h -- C ABI of librxunet.so: hand-written CDNA4 (gfx950) kernels for the hot path of the
 * multi-task 3-D residual-encoder U-Net (reference: /root/reference/builders/, path
 * NetworkFromConfig.forward + autograd backward).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator);
 *    nothing here allocates, frees, synchronises or retains pointers after return;
 *  - every entry point enqueues on `stream` (a hipStream_t passed as void*) and returns
 *    RX_OK (0) or a negative status; rx_last_error() gives a human-readable reason;
 *  - activations inside the engine are CHANNELS-LAST (n, z, y, x, c) in the compute dtype
 *    (RX_F32 parity mode, RX_BF16 / RX_F16 throughput modes); the boundary tensors (input
 *    image, logits) are NCDHW fp32 exactly as the reference's callers see them
 *    (train.py:195-204, dataset.py:211-220);
 *  - parameters and parameter gradients cross the boundary in PyTorch's own layouts
 *    (Conv3d (Co,Ci,kz,ky,kx), ConvTranspose3d (Ci,Co,kz,ky,kx)), fp32.
 *
 * Environment knobs (RX_*, listed in README.md) are measurement switches: the library reads each ONCE per thread, at the first
 * call that consults it, and keeps the answer -- changing one inside a running process has no effect.
 *
 * Each entry point names the torch primitive of the reference it replaces (file:line relative to
 * /root/reference).  The Python binding a maintainer would add is in INTEGRATION.md.
 */
#ifndef RXUNET_H
#define RXUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RX_OK 0
#define RX_EINVAL (-1)      /* bad argument (null pointer, non-positive size, misaligned) */
#define RX_EUNSUPPORTED (-2) /* shape outside what the kernels cover (e.g. channels % 32 != 0) */
#define RX_ELAUNCH (-3)     /* HIP reported a launch error */
#define RX_EWORKSPACE (-4)  /* workspace too small */

typedef enum { RX_F32 = 0, RX_BF16 = 1, RX_F16 = 2 } rx_dtype;
typedef enum { RX_ACT_NONE = 0, RX_ACT_SIGMOID = 1, RX_ACT_SOFTMAX = 2 } rx_head_act;

/* channels-last activation view: element (n,z,y,x,c) lives at
 * ptr[((n*z_*y_ + ...)*x_ + x)*ld + c]; `ld` >= c lets a view address a channel slice of a wider
 * buffer (how torch.cat of decoder.py:147 is eliminated). */
typedef struct {
  void* ptr;
  int32_t n, z, y, x, c, ld;
  /* cs == 0: the c channels of a voxel are contiguous (stride ld between voxels).
   * cs != 0 ("planar concat", only the 3x3x3 stride-1 conv entry points take it, 16-bit types, ld == 32): the channels come
   * in groups of ld, group j of element (n,z,y,x) lives at ptr[j*cs + (((n*z_+z)*y_+y)*x_+x)*ld + c%ld] -- how the two
   * 32-channel halves of the full-resolution decoder.py:147 concat stay DENSE tensors (a 64-byte channel slice of a 128-byte
   * voxel costs every kernel that streams it 1.5-2x: measured 154 -> 250 us for the stride-2 conv reading it). */
  int64_t cs;
} rx_act;

int rx_abi_version(void);
const char* rx_last_error(void);
int rx_device_arch_ok(void); /* 1 iff the current device is gfx950 */
/* name of the kernel instantiation the last conv / convT entry point of this thread launched (bench attribution) */
const char* rx_last_conv_kernel(void);

/* ---- parameter packing ------------------------------------------------------------------ */
/* Conv3d weight (Co,Ci,T) fp32 -> w_fwd [T][Co][Ci] and w_bwd [T][Ci][Co] (same tap order; the
 * bwd-data tap table mirrors the offsets instead), both in `dt`.  Either output
 * may be NULL.  T = kz*ky*kx. */
int rx_pack_conv_weight(rx_dtype dt, const float* w, int co, int ci, int taps, void* w_fwd, void* w_bwd,
                        void* stream);
/* ConvTranspose3d weight (Ci,Co,T) fp32 -> w_fwd [T][Co][Ci], w_bwd [T][Ci][Co] (tap order kept). */
int rx_pack_convT_weight(rx_dtype dt, const float* w, int ci, int co, int taps, void* w_fwd, void* w_bwd,
                         void* stream);
/* the same for `count` weights in ONE launch per 40 tensors (a train step re-packs every conv weight: 66 tensors at cfg2).
 * kind[i] 0: Conv3d weight (A = Co, B = Ci), 1: ConvTranspose3d weight (A = Ci, B = Co); HOST arrays of device pointers. */
int rx_pack_multi(rx_dtype dt, int count, const float* const* w, const int* kind, const int* A, const int* B,
                  const int* taps, void* const* w_fwd, void* const* w_bwd, void* stream);

/* ---- nn.Conv3d (simple_conv_blocks.py:43-51; kernel per axis in {1,3}, stride per axis in {1,2},
 *      padding (k-1)/2, dilation 1) ------------------------------------------------------- */
/* `ws`/`ws_bytes`: optional scratch for split-K (used only for the deep 4^3/8^3 layers whose natural
 * grid cannot fill 256 CUs); NULL disables split-K.  rx_conv_workspace_hint() is always enough. */
size_t rx_conv_workspace_hint(void);
int rx_conv3d_fwd(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                  const int32_t kernel[3], const int32_t stride[3], void* ws, size_t ws_bytes, void* stream);
/* the same plus the InstanceNorm statistics of y (stats[n][c] = (mean, rstd), as rx_instnorm_stats): on the persistent
 * halo kernels the sums come out of the conv epilogue (per-lane running sums, one wavefront reduction per workgroup) and y
 * is not read again; otherwise conv followed by rx_instnorm_stats.  ws >= max(rx_conv_workspace_hint(),
 * rx_instnorm_stats_workspace(y)). */
int rx_conv3d_fwd_stats(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                        const int32_t kernel[3], const int32_t stride[3], float eps, float* stats, void* ws,
                        size_t ws_bytes, void* stream);
/* dx (+)= conv_transpose(dy, w): autograd of the above w.r.t. its input */
int rx_conv3d_bwd_data(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx,
                       const int32_t kernel[3], const int32_t stride[3], int accumulate, void* ws,
                       size_t ws_bytes, void* stream);
/* backward-data that also delivers the two means of the InstanceNorm backward of the layer whose OUTPUT gradient it completes
 * (dx = dL/d(out of that layer); in_y / in_stats / slope describe it; the layer has no residual: its LeakyReLU mask is the
 * sign of the normalised value; the caller guarantees nothing adds to dx afterwards).  On the persistent 32-channel halo
 * kernel  sum g'  and  sum g'*(y - mean)  are per-lane running sums of the epilogue (y prefetched under the MFMA loop):
 * *fused = 1, m12[n][c] = (mean g', mean g'*xhat), continue with rx_instnorm_act_bwd_apply.  Otherwise *fused = 0 and m12 is
 * untouched: continue with rx_instnorm_act_bwd. */
int rx_conv3d_bwd_data_instats(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx,
                               const int32_t kernel[3], const int32_t stride[3], int accumulate, const rx_act* in_y,
                               const float* in_stats, float slope, float* m12, int* fused, void* ws, size_t ws_bytes,
                               void* stream);
/* dw (Co,Ci,T) fp32 = sum over voxels; workspace from rx_conv3d_bwd_weight_workspace() */
size_t rx_conv3d_bwd_weight_workspace(const rx_act* x, const rx_act* dy, const int32_t kernel[3]);
int rx_conv3d_bwd_weight(rx_dtype dt, const rx_act* x, const rx_act* dy, float* dw, const int32_t kernel[3],
                         const int32_t stride[3], void* ws, size_t ws_bytes, void* stream);

/* ---- nn.ConvTranspose3d with kernel == stride, per axis in {1,2} (decoder.py:110-113,146) -- */
int rx_convT3d_fwd(rx_dtype dt, const rx_act* x, const void* w_fwd, const float* bias, const rx_act* y,
                   const int32_t stride[3], void* ws, size_t ws_bytes, void* stream);
int rx_convT3d_bwd_data(rx_dtype dt, const rx_act* dy, const void* w_bwd, const rx_act* dx,
                        const int32_t stride[3], int accumulate, void* ws, size_t ws_bytes, void* stream);
size_t rx_convT3d_bwd_weight_workspace(const rx_act* x, const rx_act* dy, const int32_t stride[3]);
int rx_convT3d_bwd_weight(rx_dtype dt, const rx_act* x, const rx_act* dy, float* dw, const int32_t stride[3],
                          void* ws, size_t ws_bytes, void* stream);

/* ---- nn.InstanceNorm3d(affine=False, eps) + LeakyReLU(slope) + residual add
 *      (build_network_from_config.py:172,208-210; resblocks.py:106-114) ------------------- */
/* stats[n][c] = (mean, rstd) with biased variance; ws: rx_instnorm_stats_workspace() bytes */
size_t rx_instnorm_stats_workspace(const rx_act* y);
int rx_instnorm_stats(rx_dtype dt, const rx_act* y, float eps, float* stats, void* ws, size_t ws_bytes,
                      void* stream);
/* nn.Dropout3d / nn.Dropout2d (channel dropout, training mode) between a conv and its InstanceNorm
 * (simple_conv_blocks.py:57-66; build_network_from_config.py:169-170 `dropout_op_kwargs`): IN(s*y) with s = 1/(1-p) equals
 * (y - mean)/sqrt(var + eps*(1-p)^2) for a kept (n, c) plane and 0 for a dropped one -- so the caller computes the statistics
 * with eps*(1-p)^2 and this call sets rstd = 0 for the dropped planes (stats[i].rstd *= keep[i], keep[n*C+c] in {0,1}); every
 * forward and backward InstanceNorm entry point then does the right thing from `stats` alone, no pass over y. */
int rx_instnorm_stats_mask(float* stats, const float* keep, int count, void* stream);
/* out = lrelu_slope( (y-mean)*rstd + residual ); residual may be NULL; slope = 1 -> no activation */
int rx_instnorm_act_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* residual,
                        const rx_act* out, float slope, void* stream);
/* stats + apply in one call: one launch when the tensor is small (<= 8^3 voxels per sample, C % 32 == 0), otherwise the two
 * calls above.  Writes stats (kept for the backward). */
int rx_instnorm_fwd(rx_dtype dt, const rx_act* y, float eps, float* stats, const rx_act* residual,
                    const rx_act* out, float slope, void* ws, size_t ws_bytes, void* stream);
/* g = dL/dout; `out` supplies the sign for the LeakyReLU mask.  out == NULL with slope != 1 means "no residual was
 * added": the mask is then the sign of the normalised value and the output tensor is not read.
 * dy = dL/dy; d_residual (optional) receives (or accumulates) g*mask.  ws as for stats. */
int rx_instnorm_act_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out,
                        float slope, const rx_act* dy, const rx_act* d_residual, int accumulate_residual,
                        void* ws, size_t ws_bytes, void* stream);

/* rx_instnorm_act_bwd for a residual-block epilogue out = lrelu(IN(y) + res) (resblocks.py:113-114) with the masked gradient
 * g' = g * lrelu'(out) written ONCE: the reduce pass stores g' into d_residual (it IS the residual's gradient; the buffer must
 * not hold earlier contributions -- d_residual == g is allowed) while it sums, the apply pass reads (g', y) only: 7 tensor
 * passes instead of 8.  pool_dy / pool_stride (optional): gradient of the AvgPool that opens the next stage's skip path
 * (resblocks.py:95), added on the fly, g <- g + pool_dy[v / stride] / prod(stride) -- replaces a preceding
 * rx_avgpool_bwd(pool_dy, g, stride, accumulate = 1).  ws as for rx_instnorm_act_bwd. */
int rx_instnorm_act_bwd_res(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out,
                            float slope, const rx_act* pool_dy, const int32_t pool_stride[3], const rx_act* d_residual,
                            const rx_act* dy, void* ws, size_t ws_bytes, void* stream);

/* second pass of rx_instnorm_act_bwd alone, with the two means m12[n][c] supplied by the caller */
int rx_instnorm_act_bwd_apply(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out,
                              float slope, const float* m12, const rx_act* dy, const rx_act* d_residual,
                              int accumulate_residual, void* stream);

/* ---- SqueezeExcite + DropPath of the residual blocks (resblocks.py:79-87,109-112 BasicBlockD; :203-212,234-240
 *      BottleneckD).  Both classes come from the un-vendored dynamic_network_architectures package: PARITY UNPINNED
 *      (restated in oracle/resenc_oracle.py).  The block output is
 *          a = lrelu( mult[n][line][c] * xhat + residual ),   mult = path_scale[n] * gate,
 *      gate = sigmoid(fc2(relu(fc1(p)))), p = (path_scale[n] * xhat).mean((2, 3)): a 5-D tensor is pooled over (z, y) and
 *      keeps x (keep_x = 1: line = x), a 4-D tensor (2-D nets, unit z axis here) over (y, x) (keep_x = 0: one line).
 *      path_scale[n] = bernoulli(keep_prob)/keep_prob of DropPath in training, NULL otherwise. -------------------- */
typedef struct {
  const float* w1; /* fc1.weight (rd, C) */
  const float* b1; /* fc1.bias (rd) */
  const float* w2; /* fc2.weight (C, rd) */
  const float* b2; /* fc2.bias (C) */
  int32_t rd;      /* reduction channels, <= 64 */
  int32_t keep_x;
} rx_se_params;
size_t rx_se_workspace(const rx_act* y);
/* line sums of y -> pooled [n][L][c] (raw line mean of xhat), hidden [n][L][rd], gate, mult [n][L][c] (all fp32, kept for
 * the backward).  se == NULL: DropPath only, mult[n][x][c] = path_scale[n] (keep_x = 1 layout), nothing else is written. */
int rx_se_gate_fwd(rx_dtype dt, const rx_act* y, const float* stats, const float* path_scale, const rx_se_params* se,
                   float* pooled, float* hidden, float* gate, float* mult, void* ws, size_t ws_bytes, void* stream);
/* out = lrelu_slope( mult * (y-mean)*rstd + residual ) */
int rx_instnorm_gate_act_fwd(rx_dtype dt, const rx_act* y, const float* stats, const float* mult, int keep_x,
                             const rx_act* residual, const rx_act* out, float slope, void* stream);
/* first backward pass: line sums of g' = g*lrelu'(out) and g'*xhat, gate backward.  Writes dadd [n][L][c] (the pooled
 * path's contribution to dL/dxhat), m12 [n][c] (the two InstanceNorm backward means) and the fc gradients
 * (dw1 (rd,C), db1 (rd), dw2 (C,rd), db2 (C); untouched when se == NULL). */
int rx_se_gate_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out, float slope,
                   const float* path_scale, const rx_se_params* se, const float* pooled, const float* hidden,
                   const float* gate, const float* mult, float* dadd, float* m12, float* dw1, float* db1, float* dw2,
                   float* db2, void* ws, size_t ws_bytes, void* stream);
/* second pass: dy = rstd*(g'*mult + dadd - m1 - xhat*m2); d_residual (+)= g' */
int rx_instnorm_gate_act_bwd(rx_dtype dt, const rx_act* g, const rx_act* y, const float* stats, const rx_act* out,
                             float slope, const float* mult, const float* dadd, const float* m12, int keep_x,
                             const rx_act* dy, const rx_act* d_residual, int accumulate_residual, void* stream);

/* ---- nn.AvgPool3d(kernel=stride, per axis in {1,2}) (resblocks.py:95) -------------------- */
int rx_avgpool_fwd(rx_dtype dt, const rx_act* x, const rx_act* y, const int32_t stride[3], void* stream);
/* rx_instnorm_act_fwd and rx_avgpool_fwd of its output in one pass (the last block of an encoder stage feeds the AvgPool of
 * the next stage's skip path, resblocks.py:95): out as above, pooled = avgpool(out), bit-identical to the two calls */
int rx_instnorm_act_pool_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* residual,
                             const rx_act* out, const rx_act* pooled, const int32_t stride[3], float slope,
                             void* stream);
int rx_avgpool_bwd(rx_dtype dt, const rx_act* dy, const rx_act* dx, const int32_t stride[3], int accumulate,
                   void* stream);

/* ---- stem: first Conv3d on the NCDHW fp32 image, Cin <= 16 with Cout * Cin * taps * 4 <= 160 KB (encoder.py:84; MFMA kernels
 *      for Cin <= 4, VALU kernels with all weights in LDS above) ---- */
int rx_stem_conv_fwd(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x, const float* w,
                     const float* bias, const rx_act* out, const int32_t kernel[3], void* stream);
/* rx_stem_conv_fwd followed by rx_instnorm_stats of its output (workspace: rx_instnorm_stats_workspace(out)); on the MFMA
 * kernel the statistics come out of the same pass.  Same (mean, rstd) either way. */
int rx_stem_conv_fwd_stats(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x, const float* w,
                           const float* bias, const rx_act* out, const int32_t kernel[3], float eps, float* stats,
                           void* ws, size_t ws_bytes, void* stream);
size_t rx_stem_conv_bwd_weight_workspace(int cin, int cout, int taps);
int rx_stem_conv_bwd_weight(rx_dtype dt, const float* x_ncdhw, int n, int cin, int z, int y, int x,
                            const rx_act* dy, float* dw, const int32_t kernel[3], void* ws, size_t ws_bytes,
                            void* stream);

/* ---- task head: Conv3d 1x1x1 with bias to K <= 64 channels, NCDHW fp32 logits, optional
 *      eval-mode activation (decoder.py:131,151-152; build_network_from_config.py:320-323) ---- */
int rx_head_fwd(rx_dtype dt, const rx_act* x, const float* w, const float* b, int k, float* out_ncdhw,
                int act, void* stream);
size_t rx_head_bwd_workspace(const rx_act* x, int k);
int rx_head_bwd(rx_dtype dt, const float* dout_ncdhw, const rx_act* x, const float* w, int k, const rx_act* dx,
                float* dw, float* db, void* ws, size_t ws_bytes, void* stream);

/* InstanceNorm apply + LeakyReLU of the layer under a task head AND the head's 1x1x1 conv (+ eval-mode activation) in one pass:
 * the activated output is written and not re-read -- or, with out = NULL, not written at all (nobody but the head reads that
 * layer's output, and rx_instnorm_act_bwd_head rebuilds what the backward needs from y).  Same `out` bit for bit and the same
 * logits to fp32 round-off as rx_instnorm_act_fwd followed by rx_head_fwd (decoder.py:115-131,151-152).  16-bit types, k <= 4. */
int rx_instnorm_act_head_fwd(rx_dtype dt, const rx_act* y, const float* stats, const rx_act* out, float slope,
                             const float* head_w, const float* head_b, int k, float* out_ncdhw, int act, void* stream);

/* InstanceNorm + LeakyReLU backward of the layer that feeds a task head (the conv block of the last decoder stage,
 * decoder.py:115-131: no residual), with the head's data gradient formed on the fly: g[v][c] = sum_k dout[k][v] * w[k][c] is
 * never written (call rx_head_bwd with dx = NULL for dw / db).  Same dy, bit for bit, as rx_head_bwd(dx = g) followed by
 * rx_instnorm_act_bwd(g, y, stats, out = NULL, ...).  k <= 4; ws as for rx_instnorm_act_bwd.
 * head_dw (k, C) / head_db (k), optional and together (round 3): the head's OWN parameter gradients out of the same reduce pass,
 * dw[k][c] = sum dout[k][v] * lrelu(xhat)[v][c] with the activation recomputed from y and rounded as the forward stored it --
 * rx_head_bwd is then not called and rx_instnorm_act_head_fwd may be given out = NULL (the activated output of that layer, 268 MB
 * at cfg2, is neither written nor read). */
int rx_instnorm_act_bwd_head(rx_dtype dt, const float* dout_ncdhw, int k, const float* head_w, const rx_act* y,
                             const float* stats, float slope, const rx_act* dy, float* head_dw, float* head_db, void* ws,
                             size_t ws_bytes, void* stream);

/* ---- per-channel sum over (n, voxels): bias gradients ----------------------------------- */
size_t rx_channel_sum_workspace(const rx_act* x);
int rx_channel_sum(rx_dtype dt, const rx_act* x, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- task losses of the train step, single pass (reference training/losses/losses.py) ---------
 * logits / target / pred: (N, C, V) fp32, NCDHW-contiguous (V = Z*Y*X).  `loss` and `grad_loss` are DEVICE scalars
 * (no host synchronisation); `coef` carries the per-channel backward coefficients from fwd to bwd
 * (2*C floats for BCE-Dice, 1 float for masked cosine).  ws: rx_loss_workspace() bytes. */
size_t rx_loss_workspace(int n, int c, long v);
/* BCEDiceLoss(alpha, beta): alpha * BCE-with-logits on targets t*(1-2s)+s (losses.py:217-238,307-318)
 * + beta * (1 - mean_c 2 sum(p t) / max(sum(p^2)+sum(t^2), eps)), p = sigmoid(logits) (losses.py:17-43,128-138) */
int rx_bce_dice_loss_fwd(const float* logits, const float* target, int n, int c, long v, float alpha, float beta,
                         float smoothing, float eps, float* loss, float* coef, void* ws, size_t ws_bytes,
                         void* stream);
int rx_bce_dice_loss_bwd(const float* logits, const float* target, int n, int c, long v, float alpha, float beta,
                         float smoothing, const float* coef, const float* grad_loss, float* dlogits, void* stream);
/* MaskedCosineLoss (losses.py:187-215): 1 - sum(cos(pred/|pred|, t) m) / (sum(m) + 1e-8), m = |t| > 1e-6; C <= 8 */
int rx_masked_cosine_loss_fwd(const float* pred, const float* target, int n, int c, long v, float* loss, float* coef,
                              void* ws, size_t ws_bytes, void* stream);
int rx_masked_cosine_loss_bwd(const float* pred, const float* target, int n, int c, long v, const float* coef,
                              const float* grad_loss, float* dpred, void* stream);

/* ---- optimizer step fused with the weight re-pack (torch.optim.AdamW arithmetic: decoupled weight decay, bias
 *      correction; train.py:69-86 selects AdamW) -- one pass over a conv / convT weight updates p, exp_avg, exp_avg_sq
 *      and rewrites both packed copies.  `clip` is an optional DEVICE scalar multiplied into the gradient
 *      (clip_grad_norm_, train.py:227).  kind 0: Conv3d weight (Co,Ci,T); kind 1: ConvTranspose3d weight (Ci,Co,T).
 *      rx_adamw_flat: the same update for parameters that have no packed copy (stem, biases, heads). */
int rx_adamw_pack(rx_dtype dt, float* p, const float* grad, float* exp_avg, float* exp_avg_sq, const float* clip, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int step, int kind, int A, int B, int taps,
                  void* w_fwd, void* w_bwd, void* stream);
int rx_adamw_flat(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, const float* clip, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, long n, void* stream);
/* Global L2 norm of a list of fp32 gradient tensors and the clip coefficient of torch.nn.utils.clip_grad_norm_(params,
 * max_norm) (train.py:227) in two launches: out[0] = norm, out[1] = min(1, max_norm / (norm + 1e-6)); the coefficient is what
 * rx_adamw_flat(_multi) / rx_adamw_pack take as `clip`.  Deterministic (fixed summation order).  `partial`: device scratch of
 * rx_grad_norm_clip_partials(count, numel) floats.  Pointer arrays are host arrays. */
long rx_grad_norm_clip_partials(int count, const long* numel);
int rx_grad_norm_clip(int count, const float* const* grad, const long* numel, float max_norm, float* partial, long partial_len,
                      float* out, void* stream);
/* the same for `count` tensors of one param group (shared hyper-parameters and step): HOST arrays of device pointers */
int rx_adamw_flat_multi(int count, float* const* p, const float* const* grad, float* const* exp_avg,
                        float* const* exp_avg_sq, const long* numel, const float* clip, double lr, double beta1,
                        double beta2, double eps, double weight_decay, int step, void* stream);

/* ---- launch programs: the launch list of a forward / backward pass, recorded once and replayed from C ------------------
 * The reference's model call is ONE Python call (`model(x)`, train.py:204; `loss.backward()`, :224) behind which the
 * framework issues its kernels natively.  A host-language loop over ~700 entry points per step costs ~8 ms of host time;
 * a program issues the same calls from C.  Between rx_prog_begin and rx_prog_end every entry point above that takes a
 * `stream` BOTH executes as usual AND appends itself -- arguments by value, rx_act / kernel / stride copied -- to the
 * program (the recording pass is an ordinary step).  Streams are recorded by index into `streams` and substituted from
 * the table given to rx_prog_run; device pointers are recorded as they are, so a program is valid as long as the buffers
 * it was recorded on live at the same addresses.  rx_prog_run(first, last) replays commands [first, last) (last < 0: to
 * the end) and returns the first non-zero status.  With `ms` != NULL (last - first floats) every command is bracketed by
 * timing events on its stream and the streams are synchronised before returning (profiling replay; the only entry point
 * that synchronises). */
typedef struct rx_prog rx_prog;
rx_prog* rx_prog_create(void);
void rx_prog_destroy(rx_prog* p);
int rx_prog_begin(rx_prog* p, void* const* streams, int n_streams);
int rx_prog_end(rx_prog* p);
int rx_prog_len(const rx_prog* p);
const char* rx_prog_cmd_name(const rx_prog* p, int i);   /* entry point of command i */
const char* rx_prog_cmd_kernel(const rx_prog* p, int i); /* kernel instantiation it dispatched at record time ("" if not a conv) */
int rx_prog_cmd_stream(const rx_prog* p, int i);         /* index into the stream table */
int rx_prog_run(rx_prog* p, int first, int last, void* const* streams, int n_streams, float* ms);
/* numbered events for cross-stream ordering (recordable like everything else): rx_stream_wait makes `stream` wait for the
 * work captured by the last rx_event_record(slot) issued before it. */
int rx_event_new(void);
/* return a slot to the library (a plan frees its slots when it is destroyed; programs that mention the slot must be destroyed
 * first).  rx_event_slots_in_use: how many slots are handed out right now (leak checks). */
int rx_event_free(int slot);
int rx_event_slots_in_use(void);
int rx_event_record(int slot, void* stream);
int rx_stream_wait(int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RXUNET_H */
