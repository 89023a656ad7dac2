/* mi355seg.h — C-ABI of libmi355seg.so (MI355X / gfx950 only).
 *
 * The reference (taintpro98/rnd-semantic-segmentation) has no FFI: its hot path is
 * Python calling torch ops.  This header DEFINES the boundary underneath the
 * reference's Python class surface.  Every entry point names the
 * reference call it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *    unless a parameter says "host".  The caller owns every buffer, including
 *    workspaces (query sizes with the *_workspace functions).
 *  - activations are NHWC ("channels_last") bf16: x[b][h][w][c]; a GEMM row m is
 *    the pixel (b,h,w).  Logits / losses / gradients of parameters are fp32.
 *  - master weights are fp32 in torch's OIHW layout; kernels consume bf16 packed
 *    copies produced by mi_pack_*.
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and
 *    never synchronises, allocates or frees: safe under HIP graph capture.
 *  - return 0 on success, a negative MI_E* code otherwise; mi_last_error()
 *    returns a thread-local message.  Nothing throws.  Re-entrant.
 */
#ifndef MI355SEG_H
#define MI355SEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355SEG_VERSION 100

#define MI_OK 0
#define MI_EINVAL (-22)   /* bad shape / alignment / null pointer */
#define MI_ENOMEM (-12)   /* workspace too small */
#define MI_EHIP (-5)      /* HIP launch error */

/* epilogue flags of mi_conv_gemm (applied in this order on the fp32 accumulator) */
#define MI_EPI_SCALE_BIAS 1  /* v = v*scale[n] + bias[n]      FrozenBatchNorm2d, reference core/components/layers.py:18-23 */
#define MI_EPI_RESIDUAL 2    /* v += res[m][n]                 `out += identity`, reference core/components/resnet.py:110 */
#define MI_EPI_RELU 4        /* v = max(v,0)                   resnet.py:95,99,111 */
#define MI_EPI_MASK 8        /* v = msk[m][n] > 0 ? v : 0      ReLU backward of the producer of msk */
#define MI_EPI_OUT_F32 16    /* store fp32 instead of bf16 */
#define MI_EPI_ZSPLIT 32     /* fp32 store to [n/zgw][M][zgw] (ASPP tap planes) */
#define MI_EPI_WRITE_MASK 64 /* also store sign bits of the result: bit n%16 of uint16 mask_out[m][n/16] = (v > 0)   (N % 16 == 0) */
#define MI_EPI_LEAKY 256     /* with MI_EPI_RELU: LeakyReLU(alpha); with MI_EPI_BITMASK: its backward (v *= alpha where the bit is 0) */
#define MI_EPI_STATS 512     /* (mi_conv_gemm_stats only) plain bf16 store + per-tile sums of (out - pilot), (out - pilot)^2 per channel */
#define MI_EPI_BITMASK 128   /* like MI_EPI_MASK but `msk` points at such packed bits: 1/16 of the bytes of the bf16 tensor */

/* gather modes */
#define MI_GATHER_FWD 0      /* src = out*stride + tap*dil - pad            (forward conv, wgrad) */
#define MI_GATHER_DGRAD 1    /* src = (out + pad - tap*dil)/stride if exact (data gradient)       */

int mi_version(void);
const char* mi_last_error(void);

/* ---- weight packing (fp32 OIHW master -> bf16 GEMM operand) ------------------------------- */
/* wp[t][o][i] = bf16(w[o][i][t]),  t = ky*k+kx.  Operand of the forward conv. */
int mi_pack_weight_fwd(const float* w_oihw, void* wp_bf16, int O, int I, int ksize, void* stream);
/* wp[t][i][o] = bf16(w[o][i][t] * (scale ? scale[o] : 1)).  Operand of the data gradient; the
 * FrozenBN scale of the conv's output channel is folded in (scale is a constant buffer). */
int mi_pack_weight_dgrad(const float* w_oihw, const float* scale_o, void* wp_bf16, int O, int I, int ksize, void* stream);

/* Every conv weight of a module in one launch.  table (device, int64[n_desc][8]) rows:
 * {w_off, scale_off or -1, wp_off, wpt_off or -1 (skip the dgrad pack), O, I, k*k, first_block}; offsets in elements
 * into wflat / sflat / wp / wpt; a block packs 32 output x 128 input channels, total_blocks = sum ceil(O/32) * ceil(I/128); k*k <= 9. */
int mi_pack_weights_multi(const float* wflat, const float* sflat, void* wp_bf16, void* wpt_bf16,
                          const int64_t* table_dev, int n_desc, int total_blocks, void* stream);

/* ---- implicit-GEMM convolution, MFMA bf16 -> fp32 accumulate ----------------------------------
 * Replaces nn.Conv2d forward and the data-gradient half of convolution_backward for every
 * conv of reference core/components/resnet.py:22-30 (conv3x3 with dilation, conv1x1) and, through
 * the fused epilogue, the FrozenBN / ReLU / residual that follow it (resnet.py:93-113).
 *   out[b][ho][wo][n] = epi( sum_{t,c} a[b][src_h(ho,t)][src_w(wo,t)][c] * wp[t][n][c] )
 * a: [B][Ha][Wa][Ca] bf16, wp: [k*k][N][Ca] bf16, out: [B][Ho][Wo][N] bf16 (or fp32).
 * Requirements: Ca % 32 == 0 (Ca % 64 != 0: stride-1 / same-size launches without a residual or mask tile only - the 32-channel main loop), N % 8 == 0, 16-byte aligned pointers; N % 16 == 0 with MI_EPI_BITMASK / MI_EPI_WRITE_MASK
 * (one uint16 of sign bits per 16 channels), and when N % 128 == 0 those packed-bit tensors must be 16-byte aligned too
 * (a tile row's 16 mask bytes move as one access).  Violations return MI_EINVAL. */
int mi_conv_gemm(const void* a, const void* wp, void* out,
                 int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                 int ksize, int stride, int pad, int dil, int gather_mode,
                 const float* scale, const float* bias, const void* res, const void* msk, void* mask_out,
                 int flags, int zgw, float alpha, void* stream);

/* The wide-tile main loop of mi_conv_gemm (csrc/igemm_pp.hip: 320|256 x 256 tile, 8 waves in two groups that alternate
 * between MFMA and LDS/DMA phases, 4-slot LDS ring) called explicitly.  Same contract and epilogue flags, restricted to
 * stride 1, Ha == Ho, Wa == Wo, Ca % 32 == 0.  mi_conv_gemm picks it by its own cost model; this entry point exists so that
 * the two main loops can be compared on one shape in one process.  mtg: 16-row MFMA tiles per wave (8 or 10 -> 256 or 320
 * tile rows), 0 = choose. */
/* which main loop mi_conv_gemm takes for a shape: 1 = igemm_pp_kernel (wide tile), 0 = igemm_nt_kernel (measurement tools) */
int mi_conv_gemm_route(int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int flags);
int mi_conv_gemm_pp(const void* a, const void* wp, void* out,
                    int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                    int ksize, int stride, int pad, int dil, int gather_mode,
                    const float* scale, const float* bias, const void* res, const void* msk, void* mask_out,
                    int flags, int zgw, float alpha, int mtg, void* stream);

/* ---- two chained 1x1 convolutions of the bottleneck sequence in one launch (csrc/chain.hip) --------------------------
 * forward (backward = 0): reference core/components/resnet.py:105-113 of block i followed by :93-95 of block i+1,
 *   mid = relu(scale1 * (a . w_first^T) + shift1 + res)   [M][N1] bf16, sign bits -> bits1_out [M][N1/8]
 *   out = relu(scale2 * (mid . w_second^T) + shift2)      [M][N2] bf16, sign bits -> bits2_out [M][N2/8]
 * i.e. mi_conv_gemm(a, w_first, flags 1|2|4|64) then mi_conv_gemm(mid, w_second, flags 1|4|64), with `mid` written once and
 * never read back.  backward = 1: the data gradients of the same two convs in the order backward meets them (conv1 of block i,
 * conv3 of block i-1; weights packed by mi_pack_weight_dgrad),
 *   mid = ((a . w_first^T) + res) where bit of bits1 else 0;   out = (mid . w_second^T) where bit of bits2 else 0
 * i.e. flags 2|128 then 128.  a [M][K1], w_first [N1][K1], res / mid [M][N1], w_second [N2][N1], out [M][N2], all bf16 and
 * 16-byte aligned; the sign-bit tensors as MI_EPI_WRITE_MASK / MI_EPI_BITMASK define them.  Built for K1 = 256, N1 = 1024, N2 = 256
 * (layer3); other sizes return MI_EINVAL.  grid: workgroups (0 = one per CU). */
int mi_conv_chain(const void* a, const void* w_first, const void* res, void* mid, const void* w_second, void* out,
                  long M, int K1, int N1, int N2,
                  const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                  const void* bits1, const void* bits2, void* bits1_out, void* bits2_out,
                  int backward, int grid, void* stream);

/* ---- weight gradient (contraction over pixels), split-K with deterministic reduction ------------
 * Replaces the weight-gradient half of convolution_backward.
 *   dw[o][i][t] (+)= scale[o] * sum_m dy[m][o] * x[src(m,t)][i]
 * dy: [B][Ho][Wo][O] bf16, x: [B][Ha][Wa][I] bf16, dw: fp32 OIHW.  O % 8 == 0, I % 8 == 0.
 * out_map 0: OIHW as above (ncls ignored).  out_map 1 (ASPP): ksize must be 1 and row o = (br*9+tap)*ncls+cls of the
 * [O][I] product (O >= 36*ncls, 36*ncls <= MI_ASPP_KPAD: the column layout mi_aspp_im2col writes) is scattered to
 * dw[br][cls][i][tap] of the 4 stacked [ncls][I][3][3] tensors.  dw_elems = number of floats dw points at; the call
 * fails with MI_EINVAL instead of writing past it. */
size_t mi_conv_wgrad_workspace(int B, int Ho, int Wo, int O, int I, int ksize);
/* which kernel mi_conv_wgrad launches for a shape (measurement tools): 0 wgrad_tn_kernel (128 x 128 tile per tap), 1 wgrad_tn256_kernel
 * (opt-in), 2 wgrad_p3_kernel (opt-in), 3 wgrad_q3_kernel (3x3 stride 1: the three taps of a kernel row fused, 64 x 128 tile),
 * 4 wgrad_s4_kernel (1x1 stride 1: 32-pixel stages, three in flight) */
int mi_conv_wgrad_route(int B, int Ha, int Wa, int I, int Ho, int Wo, int O, int ksize, int stride, int pad, int dil, int out_map);
int mi_conv_wgrad(const void* dy, const void* x, float* dw,
                  int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                  int ksize, int stride, int pad, int dil,
                  const float* scale_o, int accumulate, int out_map, int ncls, size_t dw_elems,
                  void* workspace, size_t workspace_bytes, void* stream);

/* mi_conv_wgrad in two parts (round 5): _partial runs the main kernel only - the split-K slabs stay in `workspace`, which the caller keeps alive and
 * untouched until the reduction - and writes the reducer's arguments to `job` (mi_conv_wgrad_job_bytes() bytes of host memory); mi_conv_wgrad_reduce runs
 * the reducers of up to 8 consecutive job records as ONE launch (the three weight gradients of a bottleneck: one launch instead of three on the
 * weight-gradient stream).  Same fixed summation order per conv, hence the same bits as mi_conv_wgrad - with one exception: this form is the one that runs
 * BESIDE a data-gradient chain, and the fused-row 3x3 route plans its K split for that (448 instead of 512 workgroup slots, MI_WGRAD_Q3_SLOTS_BESIDE:
 * fewer, longer splits = another partition of the fp32 sums; deterministic, and within fp32 rounding of the one-call result). */
size_t mi_conv_wgrad_job_bytes(void);
int mi_conv_wgrad_partial(const void* dy, const void* x, float* dw,
                          int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                          int ksize, int stride, int pad, int dil,
                          const float* scale_o, int accumulate, int out_map, int ncls, size_t dw_elems,
                          void* workspace, size_t workspace_bytes, void* job, void* stream);
int mi_conv_wgrad_reduce(const void* jobs, int n, void* stream);

/* ---- ASPP head (reference core/models/classifiers/aspp/classifier.py:6-32) ---------------------
 * forward:  Z = X * Wall^T as ONE plain GEMM (mi_conv_gemm, ksize 1, MI_EPI_ZSPLIT, zgw 20) over all
 * 4 rates x 9 taps x 19 classes, then mi_aspp_col2im sums the 36 shifted planes (+ the 4 biases):
 *   low[b][h][w][n] = sum_r bias[r][n] + sum_{r,ky,kx} Z[r*9+ky*3+kx][b][h+(ky-1)d_r][w+(kx-1)d_r][n]
 * which is exactly out = conv_0(x); out += conv_i(x) (classifier.py:27-29) re-associated.
 * backward: mi_aspp_im2col builds G[m][(r*9+t)*19+n] = dlow[m - shift][n]; then dX = G * Wall (mi_conv_gemm),
 * dWall = G^T X (mi_conv_wgrad, out_map 1), dbias = column sums of dlow. */
#define MI_ASPP_NRATES 4
#define MI_ASPP_ZGW 20          /* classes padded 19 -> 20 per tap plane */
#define MI_ASPP_KPAD 704        /* 36*19 = 684 padded to a multiple of 64 */
/* wall[(g*20+n)][c] = bf16(w[r][n][c][tap]), g = r*9+tap; row n==19 of each plane is zero.  [720][C] */
int mi_aspp_pack_fwd(const float* w4 /*[4][K][C][3][3]*/, void* wall_bf16, int C, int K, void* stream);
/* wallT[c][g*K+n] = bf16(w[r][n][c][tap]); columns >= 36*K are zero.  [C][704] */
int mi_aspp_pack_dgrad(const float* w4, void* wallT_bf16, int C, int K, void* stream);
int mi_aspp_col2im(const float* z /*[36][M][20]*/, const float* bias4 /*[4][K]*/, float* low /*[B][H][W][K]*/,
                   int B, int H, int W, int K, const int* rates4 /*host*/, void* stream);
/* g[m][k] bf16, k = (r*9+t)*K+n, columns >= 36*K zero */
int mi_aspp_im2col(const float* dlow /*[B][H][W][K]*/, void* g_bf16 /*[M][704]*/,
                   int B, int H, int W, int K, const int* rates4 /*host*/, void* stream);
/* dbias4[r][n] (+)= sum_m dlow[m][n], r = 0..3 (every branch sees the same gradient); two-level fixed-order column sum,
 * workspace >= mi_colsum_workspace(M, K) bytes (csrc/colsum.hip) */
size_t mi_colsum_workspace(int M, int N);
int mi_aspp_bias_grad(const float* dlow, float* dbias4 /*[4][K]*/, int M, int K, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- bilinear upsample, align_corners=True (classifier.py:31, core/utils/utility.py:185) --------
 * low: [B][h][w][K] fp32 NHWC, up: [B][K][H][W] fp32 NCHW (the layout the reference returns). */
int mi_upsample_ac_fwd(const float* low, float* up, int B, int h, int w, int K, int H, int W, void* stream);
int mi_upsample_ac_bwd(const float* dup, float* dlow, int B, int h, int w, int K, int H, int W, void* stream);

/* ---- per-pixel softmax cross-entropy, ignore_index (core/trainers/aspp_trainer.py:61,91) --------
 * logits [B][K][H][W] fp32 NCHW, labels [B][H][W] int64.  loss_out is FOUR floats: [0] = mean over valid pixels
 * (nan if none), [1] = number of valid pixels, [2] = number of labels outside [0,K) that are not ignore_index (torch
 * raises a device assert for those; here they are left out of the loss and reported - a non-zero count means the
 * label map is wrong), [3] = scratch.  workspace: mi_ce_workspace bytes. */
size_t mi_ce_workspace(int B, int H, int W);
int mi_softmax_ce_fwd(const float* logits, const int64_t* labels, float* loss_out /*[4]*/,
                      int B, int K, int H, int W, int ignore_index, void* workspace, size_t workspace_bytes, void* stream);
/* dlogits = (softmax - onehot) * valid / n_valid * grad_scale, n_valid read from loss_out[1] on device */
int mi_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* loss_out, float* dlogits,
                      int B, int K, int H, int W, int ignore_index, float grad_scale, void* stream);

/* ---- fused upsample + cross-entropy: never materialises the [B][K][H][W] tensor (training path) --
 * classifier(feat, size) + CrossEntropyLoss of reference aspp_trainer.py:89-91 in one call.
 * low [B][h][w][K] fp32 NHWC, labels [B][H][W] int64 (H >= h, W >= w, K <= 32).
 * loss_out (four floats, as mi_softmax_ce_fwd): [0] = mean loss over valid pixels, [1] = n_valid, [2] = out-of-range labels.
 * dlow (may be NULL: loss only) [B][h][w][K] = d loss / d low * grad_scale, already divided by n_valid.
 * Deterministic (fixed summation order, no atomics). */
size_t mi_upsample_ce_workspace(int B, int h, int w, int K, int H, int W);
int mi_upsample_ce(const float* low, const int64_t* labels, float* loss_out /*[4]*/, float* dlow,
                   int B, int h, int w, int K, int H, int W, int ignore_index, float grad_scale,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The same with either bilinear convention: align_corners != 0 as above; 0: F.interpolate(..., size=, mode="bilinear") at its default
 * align_corners=False (source index max(in / out * (dst + 0.5) - 0.5, 0)) - the four deep-supervision heads of the GALD path
 * (core/models/classifiers/gcpacc/gcpa_cc2.py:78-81 + core/trainers/gald_trainer.py:66-84). */
int mi_upsample_ce_ex(const float* low, const int64_t* labels, float* loss_out /*[4]*/, float* dlow,
                      int B, int h, int w, int K, int H, int W, int ignore_index, float grad_scale, int align_corners,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- inference tail: upsample to label size + softmax over classes (utility.py:185-186) ---------
 * probs [B][K][H][W] fp32; pred (optional, may be NULL) [B][H][W] uint8 argmax (first max wins). */
int mi_upsample_softmax(const float* low, float* probs, uint8_t* pred, int B, int h, int w, int K, int H, int W, void* stream);

/* ---- optimiser: torch.optim.SGD(momentum, weight_decay) on flat fp32 buffers (aspp_trainer.py:25-26,94-95)
 * g' = g + wd*p; buf = mu*buf + g'; p -= lr*buf  (buf zero-initialised == torch's first-step buf = g'). */
int mi_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, void* stream);
/* the same update with hyper = {lr, momentum, weight_decay} read from DEVICE memory at run time: under HIP graph capture a
 * by-value argument is frozen into the graph, a pointer is not - the poly learning rate changes every step. */
int mi_sgd_step_dev(float* p, const float* g, float* buf, size_t n, const float* hyper /*[3], device*/, void* stream);

/* ---- stem tail: FrozenBN + ReLU + 3x3/stride 2/pad 1 max-pool fused (resnet.py:138-141) ----------------------
 * y: conv1 output [B][Hc][Wc][C] bf16 NHWC; pool [B][Hp][Wp][C] bf16; idx one byte per pooled element: the winning
 * tap 0..8 (first maximum in scan order, like ATen) or 9 when the pooled value is 0 (no gradient through ReLU).
 * backward: dy[b][h][w][c] = scale[c] * sum of dpool over the windows whose idx points at (h,w).  C % 8 == 0. */
int mi_stem_pool_fwd(const void* y, const float* scale, const float* shift, void* pool, uint8_t* idx,
                     int B, int Hc, int Wc, int C, int Hp, int Wp, void* stream);
int mi_stem_pool_bwd(const void* dpool, const uint8_t* idx, const float* scale, void* dy,
                     int B, int Hc, int Wc, int C, int Hp, int Wp, void* stream);
/* Patch matrix of the stem conv (7x7, stride 2, pad 3, 3 input channels; resnet.py:137): col[m][k] bf16 [B*Ho*Wo][ncols], k = (c*7+ky)*7+kx for
 * k < 147, zero above (ncols a multiple of 8; 192 when the matrix also feeds the forward as a 1x1 mi_conv_gemm, 160 for the weight gradient
 * alone).  With it the stem conv is a plain GEMM and its weight gradient a 1x1 mi_conv_wgrad (deterministic), not the library's atomics. */
int mi_stem_im2col(const void* x_bf16_nhwc, void* col_bf16, int B, int H, int W, int Ho, int Wo, int ncols, void* stream);

/* ---- FADA adversarial step (reference core/combos/aspp_fada.py:80-127) ---------------------------
 * db[n] (+)= sum_m dy[m][n]  (conv bias gradient; dy bf16 [M][N], N % 8 == 0; fixed summation order) */
int mi_bias_grad_bf16(const void* dy_bf16, float* db, int M, int N, int accumulate,
                      void* workspace /* mi_colsum_workspace(M, N) */, size_t workspace_bytes, void* stream);
/* Fused  soft = clip(softmax(upsample(seg_low) * inv_temperature), clip);  pred = upsample(d_low)[:, :2K];
 *        loss = mean_pixels( -sum_c soft_c * log_softmax(pred)[c + domain*K] )
 * = soft_label_cross_entropy(model_D(fea, size), cat(soft, 0) or cat(0, soft)) of aspp_fada.py:104-121 with the soft labels
 * of :96-103, without materialising any [B,C,H,W] tensor.  seg_low [B][h][w][K], d_low [B][h][w][ldD] (first 2K channels
 * used), dd_low (may be NULL) = d loss / d d_low * grad_scale, same layout (padding channels zeroed).  loss_out[0] = loss,
 * loss_out[1] = pixel count.  K <= 32, 2K <= ldD. */
size_t mi_upsample_softce_workspace(int B, int h, int w, int K, int H, int W);
int mi_upsample_softce(const float* seg_low, float inv_temperature, float clip, const float* d_low, int ldD, int domain,
                       float* loss_out, float* dd_low, int B, int h, int w, int K, int H, int W, float grad_scale,
                       void* workspace, size_t workspace_bytes, void* stream);
/* torch.optim.Adam (no amsgrad, weight_decay 0; fada_adapter.py:24) on flat fp32 buffers; step >= 1 is this update's index */
int mi_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                 float eps, int step, void* stream);
/* the same after clamping g to [-grad_clamp, grad_clamp] in place: clip_gradient (core/utils/utils.py:6-16) + Adam.step of pranet_trainer.py:59-60 */
int mi_adam_step_clamped(float* p, float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                         float eps, int step, float grad_clamp, void* stream);

/* the same with hyper = {lr, beta1, beta2, eps, grad_clamp (<= 0: none), step} read from DEVICE memory at run time (HIP-graph replay: by-value
 * arguments are frozen into a captured graph; the step count and with it the bias corrections change on every replay) */
int mi_adam_step_dev(float* p, float* g, float* exp_avg, float* exp_avg_sq, size_t n, const float* hyper /*[6], device*/, void* stream);

/* ---- exact-fp32 evaluation path (test.py / ASPPTester; csrc/igemm_f32.hip) ---------------------------------------
 * The reference computes in fp32; BASELINE.json asks for argmax masks identical to it.  These entry points run the same
 * graph on fp32 NHWC activations with fp32 weights on the f32-input MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain,
 * one rounding per product).  Forward only.
 * mi_pack_weight_f32: wp[t][o][i] = w[o][i][t] (no rounding; for ksize 1 the OIHW tensor itself is already this layout).
 * mi_conv_f32: out[b][ho][wo][n] = epi( sum_{t,c} a[b][ho*stride+ky*dil-pad][wo*stride+kx*dil-pad][c] * wp[t][n][c] ),
 *   a [B][Ha][Wa][Ca] fp32 (Ca % 16 == 0), out [B][Ho][Wo][N] fp32 (any N).  flags: MI_EPI_SCALE_BIAS (v*scale[n] then
 *   +bias[n], two rounded operations like layers.py:21-23; scale NULL = bias only, nn.Conv2d(bias=True) of classifier.py:12-20),
 *   MI_EPI_RESIDUAL (v += res[m][n]: `out += identity` resnet.py:110, or `out += conv_i(x)` classifier.py:28-29), MI_EPI_RELU.
 * mi_stem_f32: x [B][3][H][W] fp32 NCHW (the loader's layout) -> relu(bn(conv 7x7/2/3)) as [B][Hc][Wc][64] fp32 NHWC
 *   (resnet.py:137-139), w = conv1.weight [64][3][7][7], scale/shift = FrozenBN fold.
 * mi_maxpool_f32: 3x3 / stride 2 / pad 1 max-pool on fp32 NHWC (resnet.py:141), C % 4 == 0. */
int mi_pack_weight_f32(const float* w_oihw, float* wp, int O, int I, int ksize, void* stream);
int mi_conv_f32(const float* a, const float* wp, float* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                int ksize, int stride, int pad, int dil, const float* scale, const float* bias, const float* res,
                int flags, void* stream);
int mi_stem_f32(const float* x_nchw, const float* w, const float* scale, const float* shift, float* y_nhwc,
                int B, int H, int W, void* stream);
int mi_maxpool_f32(const float* y_nhwc, float* pool_nhwc, int B, int Hc, int Wc, int C, void* stream);

/* ---- data-parallel gradient exchange over RCCL (csrc/comm.hip) ----------------------------------------------
 * For hosts that do not go through torch.distributed (the Python product does: host/ddp.py, backend "nccl" = RCCL).  One process
 * per GPU; rank 0 calls mi_comm_unique_id and passes the 128 bytes to the other ranks by its own means; every rank calls
 * mi_comm_init_rank; per step each bucket of the flat fp32 gradient buffer is averaged in place with mi_allreduce_bucket on a
 * side stream (order it after the bucket's weight-gradient launches with an event) and the optimizer stream waits for that
 * stream before mi_sgd_step.  librccl.so is resolved at the first call (dlopen; a copy already in the process is reused).
 * dtype: 0 = fp32, 1 = bf16; average != 0 divides by the number of ranks (ncclAvg). */
int mi_comm_unique_id(void* id128 /* host, 128 bytes out */);
int mi_comm_init_rank(void** comm /* out */, int nranks, const void* id128 /* host */, int rank);
int mi_comm_destroy(void* comm);
int mi_allreduce_bucket(void* ptr, size_t count, int dtype, int average, void* comm, void* stream);

/* ---- elementwise helpers ------------------------------------------------------------------------ */
/* y = msk > 0 ? x : 0 (bf16, n % 8 == 0): ReLU backward across an autograd boundary.  bits != 0: `msk` is the packed
 * sign-bit tensor written by MI_EPI_WRITE_MASK (same element order). */
int mi_relu_mask(const void* x_bf16, const void* msk, void* y_bf16, size_t n, int bits, void* stream);
/* FrozenBN fold: scale = w*rsqrt(var) (no eps), shift = b - mean*scale (layers.py:18-20) */
int mi_frozen_bn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift, int n, void* stream);

/* ---- trainable BatchNorm2d (MODEL.FREEZE_BN=False: feature_extractor.py:37-39 builds the backbone on torch.nn.BatchNorm2d) over NHWC
 * bf16 activations [M][C], C % 8 == 0, C <= 2048 for the reductions.  Reductions return RAW per-channel sums in a fixed order (bitwise
 * reproducible); the caller divides by the pixel count, after an all-reduce over ranks when the statistics are synchronised
 * (train_distill.py:53 converts to SyncBatchNorm).  workspace: mi_bn_workspace(M, C) bytes. */
size_t mi_bn_workspace(long M, int C);
/* out[c] = sum_m y[m][c] (mean == NULL) or sum_m (y[m][c] - mean[c])^2 (two-pass variance) */
int mi_bn_colsum(const void* y_bf16, const float* mean, long M, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* one pass: s1[c] = sum_m (y[m][c] - pilot[c]), s2[c] = sum_m (y[m][c] - pilot[c])^2; mean = pilot + s1/N, var = s2/N - (s1/N)^2.  The pilot
 * must be the same on every rank that shares the statistics (the running mean is). */
int mi_bn_colsum2(const void* y_bf16, const float* pilot, long M, int C, float* s1, float* s2, void* workspace, size_t workspace_bytes, void* stream);
/* Conv forward (mi_conv_gemm's contract, plain bf16 store) that also returns what mi_bn_colsum2 would compute on its output: sums[0][n] = sum_m (out[m][n] -
 * pilot[n]), sums[1][n] = sum_m (out[m][n] - pilot[n])^2 over the bf16-rounded outputs, accumulated per row tile in the epilogue and added in a fixed
 * order (bitwise reproducible) - the statistics pass over the conv output of `bn(conv(x))` (resnet.py:93-109 with feature_extractor.py:37) costs no
 * extra read.  gamma != NULL: the last reduction launch also does mi_bn_finalize's work with count = B*Ho*Wo (gamma, beta, running_mean / running_var or
 * both NULL, num_batches_tracked or NULL, momentum, eps, out4 [4][N]) - the unsynchronised BatchNorm2d; gamma == out4 == NULL: sums only (the caller
 * all-reduces them and calls mi_bn_finalize).  workspace: mi_conv_gemm_stats_workspace(B*Ho*Wo, N) bytes. */
size_t mi_conv_gemm_stats_workspace(long M, int N);
int mi_conv_gemm_stats(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int pad,
                       int dil, const float* pilot, float* sums, void* workspace, size_t workspace_bytes, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* out4, void* stream);
/* (s1, s2) of mi_bn_colsum2 over `count` pixels (all ranks) -> out4 = [mean | invstd | gamma * invstd | beta - mean * gamma * invstd] ([4][C] fp32), in
 * double; running_mean / running_var (both or neither) updated as torch.nn.BatchNorm2d does (momentum, unbiased variance), num_batches_tracked
 * (optional, int64 scalar) += 1.  pilot may alias running_mean (it is read first).  Replaces the host arithmetic between statistics and normalise pass
 * of resnet.py:84-113's BatchNorm2d in train(). */
int mi_bn_finalize(const float* s1, const float* s2, const float* pilot, double count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float momentum, float eps, float* out4, int C, void* stream);
/* out = relu?((y - mean[c]) * scale[c] + beta[c] (+ res)), scale = gamma * rsqrt(var + eps); mask_out (optional, C % 16 == 0): the
 * packed sign bits of out in the layout of MI_EPI_WRITE_MASK */
int mi_bn_apply(const void* y_bf16, const float* mean, const float* scale, const float* beta, const void* res_bf16, void* out_bf16,
                void* mask_out, int relu, long M, int C, void* stream);
/* dbeta[c] = sum_m g[m][c], dgamma[c] = sum_m g[m][c] * (y[m][c] - mean[c]) * invstd[c]   (raw sums); relu_bits (optional, C % 16 == 0):
 * the packed sign bits of the layer's output - g counts only where the bit is set (ReLU backward without a separate pass) */
int mi_bn_bwd_colsums(const void* g_bf16, const void* y_bf16, const float* mean, const float* invstd, const void* relu_bits, long M, int C,
                      float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes, void* stream);
/* dy = gamma * invstd * (g - dbeta * inv_count - xhat * dgamma * inv_count), xhat = (y - mean) * invstd; inv_count = 1 / (pixels over all
 * ranks that shared the statistics) */
int mi_bn_bwd_apply(const void* g_bf16, const void* y_bf16, const float* mean, const float* invstd, const float* gamma, const float* dbeta,
                    const float* dgamma, float inv_count, const void* relu_bits, void* dy_bf16, long M, int C, void* stream);

/* ---- PraNet path, first kernel: the structure loss of pranet_trainer.py:22-31 on one-channel maps pred / mask fp32
 * [B][H][W]: weit = 1 + 5 |avg_pool31(mask) - mask|, BCE-with-logits averaged over the whole batch (the reference's reduce='none' is read by
 * torch as the legacy reduce=True), weighted IoU per image.  out: float[1 + 2B] = {loss, N_0, D_0, ...} (N = inter + 1, D = union - inter + 1);
 * grad (optional) = grad_scale * d loss / d pred.  Fixed-order sums. */
size_t mi_structure_loss_workspace(int B, int H, int W);
int mi_structure_loss(const float* pred, const float* mask, int B, int H, int W, float* out, float* grad, float grad_scale,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- PraNet path, the network (csrc/gconv.hip, csrc/gnet.hip) ---------------------------------------------------
 * Operands are channel-slice VIEWS of NHWC tensors: a pointer to channel 0 of the slice in pixel 0 and `ld`, the elements per pixel
 * row of the tensor the slice lives in (ld >= channels).  torch.split / torch.cat of the Res2Net bottleneck (Res2Net_v1b.py:68-84),
 * of RFB_modified (PraNet_Res2Net.py:50-54) and of the partial decoder (:84-91) therefore never copy.  Any channel count; alignment
 * only selects the access width (16 / 4 / 2 bytes).  bf16 unless a parameter says fp32.
 *
 * mi_gconv: nn.Conv2d of every shape the path has (kh x kw taps, per-axis stride / padding / dilation, optional bias) and, with
 *   MI_GATHER_DGRAD on the transposed pack, its data gradient.  a [B][Ha][Wa][Ca], out [B][Ho][Wo][N] (the OUTPUT of the launch:
 *   in DGRAD mode that is the forward conv's input shape and a is the forward conv's output gradient).
 *   wp: [kh*kw][roundup(N,32)][roundup(Ca,32)] bf16 from mi_gconv_pack_multi (zero padded).
 *   stats (optional): float[ceil(M/128)][2][N], per 128-pixel tile the column sums and sums of squares of the bf16-rounded outputs -
 *   the first level of nn.BatchNorm2d's batch statistics (PraNet_Res2Net.py:13), finished by mi_gbn_finalize.
 *   out_f32: store fp32 (the one-channel side maps; N <= 32). */
size_t mi_gconv_pack_elems(int O, int I, int kh, int kw);
/* table (device, int64[n_desc][8]) rows: {w_off, wp_off, wpt_off or -1, O, I, kh*kw, first_block, 0}; w_off indexes wflat (fp32 OIHW
 * masters), wp_off / wpt_off the packed forward [t][Opad][Ipad] and data-gradient [t][Ipad][Opad] operands; a block packs 1024 elements,
 * total_blocks = sum ceil(kh*kw*Opad*Ipad / 1024). */
int mi_gconv_pack_multi(const float* wflat, void* wp_bf16, void* wpt_bf16, const int64_t* table_dev, int n_desc, int total_blocks, void* stream);
size_t mi_gconv_stats_elems(int B, int Ho, int Wo, int N);
int mi_gconv(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
             int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int mode, const float* bias, float* stats, int out_f32,
             void* stream);
/* dw[o][i][ky][kx] (+)= sum_m dy[m][o] * x[src(m,ky,kx)][i], fp32 OIHW; split-K slabs summed in a fixed order. */
size_t mi_gconv_wgrad_workspace(int B, int Ho, int Wo, int O, int I, int kh, int kw);
int mi_gconv_wgrad(const void* dy, long ldy, const void* x, long ldx, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw_, int accumulate, void* workspace, size_t workspace_bytes, unsigned* tickets, int n_tickets,
                   void* stream);      /* tickets: NULL, or n_tickets zeroed words (left zero): with few K splits the slabs are then added inside the first launch (same bits) */
/* The weight gradients of many convs in one go (the tape of host/pranet.py queues them and flushes once per backward: nothing reads a weight gradient
 * before the optimizer; reference: the autograd of nn.Conv2d.weight, PraNet_Res2Net.py:7-20, hardnet_68.py:56-80).  Each job is mi_gconv_wgrad's argument
 * list; two jobs must not name the same dw.  table_dev: device scratch of mi_gconv_wgrad_multi_table_bytes(n) bytes; workspace: device scratch of
 * mi_gconv_wgrad_multi_workspace(jobs, n) bytes (fp32 split-K slabs of all jobs).  Launches: the table writers (descriptors travel as kernel arguments),
 * one main kernel per operand-alignment class, one reducer.  The K split per job is ~48 steps of 64 pixels (MI_GWM_STEPS) - not mi_gconv_wgrad's split, so the
 * fp32 summation order (not the result beyond rounding) differs from the one-conv call. */
typedef struct MiWgradJob {
    const void* dy; long ldy; const void* x; long ldx; float* dw;
    int B, Ha, Wa, I, Ho, Wo, O, kh, kw, sh, sw, ph, pw, dh, dw_, accumulate;
} MiWgradJob;
size_t mi_gconv_wgrad_multi_table_bytes(int n);
size_t mi_gconv_wgrad_multi_workspace(const MiWgradJob* jobs, int n);
int mi_gconv_wgrad_multi(const MiWgradJob* jobs, int n, void* table_dev, size_t table_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* nn.BatchNorm2d in train() from the conv's tile statistics: mean, invstd = rsqrt(biased var + eps), scale = gamma * invstd,
 * shift = beta - mean * scale, and the running-statistics update (momentum; unbiased variance), all per channel.  count = pixels. */
int mi_gbn_finalize(const float* partials, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out, float* shift_out, void* stream);
/* eval(): scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale */
int mi_gbn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale, float* shift, int C,
                void* stream);
/* out = act(y * scale[c] + shift[c] (+ add)); relu: 0 none, 1 ReLU, 2 ReLU6 (hardnet_68.py:78); out bf16, or fp32 when out_f32 */
int mi_gbn_apply(const void* y, long ldy, const float* scale, const float* shift, const void* add, long ldadd, void* out, long ldo, int out_f32, long M, int C,
                 int relu, void* stream);
/* mi_gbn_apply (bf16 output) with up to 4 extra destinations (round 5): channels [c0[k], c1[k]) of the result AS STORED (rounded to bf16), plus add2[k] when
 * that is not NULL, also go to the view dst[k] (ldd[k] elements per pixel row; add2[k] likewise a view of c1 - c0 channels).  Replaces the stand-alone copies of
 * a HarDBlock's gathers (reference hardnet_68.py:137-160), the pass-through group of a Res2Net bottleneck and its `sp = sp + spx[i]` (Res2Net_v1b.py:72-84)
 * with bit-identical results.  `out` may be NULL when only the extra destinations are wanted.  The arrays are HOST memory (read during the call). */
int mi_gbn_apply_multi(const void* y, long ldy, const float* scale, const float* shift, const void* add, long ldadd, void* out, long ldo, long M, int C, int relu,
                       int n_extra, const int* c0, const int* c1, void* const* dst, const long* ldd, const void* const* add2, const long* lda2, void* stream);
/* dbeta[c] (+)= sum_m g'[m][c], dgamma[c] (+)= sum_m g'[m][c] * (y[m][c] - mean[c]) * invstd[c]; g' = g where mask > 0 (mask: the layer's
 * ReLU output, or NULL); y NULL: dbeta only (a conv bias gradient).  g fp32 when g_f32; mask_f32 is a flag word: bit 0 = the mask is fp32,
 * bit 1 = the mask is a ReLU6 output (the gradient passes where 0 < mask < 6).  The same flags in mi_gbn_bwd_apply. */
size_t mi_gcolsum_workspace(long M, int C);
int mi_gbn_bwd_sums(const void* g, long ldg, int g_f32, const void* y, long ldy, const void* mask, long ldm, int mask_f32, const float* mean,
                    const float* invstd, long M, int C, float* dbeta, float* dgamma, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* dy = gamma * invstd * (g' - dbeta * inv_count - xhat * dgamma * inv_count) */
int mi_gbn_bwd_apply(const void* g, long ldg, int g_f32, const void* y, long ldy, const void* mask, long ldm, int mask_f32, const float* mean,
                     const float* invstd, const float* gamma, const float* dbeta, const float* dgamma, float inv_count, void* dy, long lddy, long M, int C,
                     void* stream);
/* out = a op b on views.  op: 0 add (sp + spx[i], Res2Net_v1b.py:72; x + crop, PraNet_Res2Net.py:140), 1 mul (partial decoder, :81-83),
 * 2 copy, 3 a where b > 0 else 0 (ReLU backward), 4 relu(a * b) (the gated products of FAM, gcpa_gald.py:88-101; bf16 only).
 * dtype: 0 bf16, 1 fp32, 2 copy fp32 -> bf16, 3 copy bf16 -> fp32. */
int mi_gbinary(int op, int dtype, const void* a, long lda, const void* b, long ldb, void* out, long ldo, long M, int C, void* stream);
/* AvgPool2d(k, stride, pad) with count_include_pad=True (include_pad != 0; Res2Net_v1b.py:40) or AvgPool2d(stride, stride, ceil_mode=True,
 * count_include_pad=False) (include_pad == 0; Res2Net_v1b.py:122-123).  backward != 0: x is d loss / d input (written), out is d loss / d output. */
int mi_gavgpool(const void* x, long ldx, void* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int include_pad,
                int backward, void* stream);
/* MaxPool2d(k, stride, pad) on NHWC bf16 views (hardnet_68.py:213,233); idx: one byte per output element, the winning tap (first maximum in
 * scan order).  backward != 0: x is d loss / d input (written), out is d loss / d output, idx as written by the forward. */
int mi_gmaxpool(const void* x, long ldx, void* out, long ldo, uint8_t* idx, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int backward,
                void* stream);
/* CrossEntropyLoss(ignore_index) on NHWC fp32 logits [M][K] (view, K <= 32), labels int64 [M] (gald_trainer.py:66-84).  loss_out: four floats as
 * mi_softmax_ce_fwd (mean over valid pixels, n_valid, out-of-range labels, -).  dlogits (optional, same layout) = d loss / d logits * grad_scale. */
size_t mi_gce_workspace(long M);
int mi_gce(const float* logits, long ld, const int64_t* labels, long M, int K, int ignore_index, float* loss_out, float* dlogits, long ldd, float grad_scale,
           void* workspace, size_t workspace_bytes, void* stream);
/* F.interpolate / nn.Upsample, mode='bilinear' (PraNet_Res2Net.py:67,127-177): src = align_corners ? scale * dst : max(scale * (dst + 0.5) - 0.5, 0).
 * scale_h / scale_w as ATen computes them: align_corners: (in - 1) / (out - 1); otherwise 1 / scale_factor when the caller gave a
 * scale_factor, in / out when it gave a size.  backward != 0: x is d loss / d input (written), out is d loss / d output (gather form,
 * fixed summation order). */
int mi_gresize(const void* x, long ldx, void* out, long ldo, int f32, int B, int H, int W, int C, int Ho, int Wo, int align_corners, float scale_h, float scale_w,
               int backward, void* stream);
/* reverse attention (PraNet_Res2Net.py:131-133): out[m][c] = (1 - sigmoid(gate[m])) * feat[m][c], gate fp32 [M]; backward: dfeat and dgate */
int mi_gra_fwd(const float* gate, const void* feat, long ldfeat, void* out, long ldo, long M, int C, void* stream);
int mi_gra_bwd(const float* gate, const void* feat, long ldfeat, const void* dy, long lddy, void* dfeat, long lddf, float* dgate, long M, int C, void* stream);

/* ---- GALD / GCPA path, first kernels (csrc/gald.hip) -----------------------------------------------------------------
 * Depthwise 3x3 convolution with bias on NHWC bf16 views: Conv2d(C, C, 3, groups=C, stride, padding) of LocalAttenModule
 * (core/models/classifiers/gcpacc/contextagg/GALDNet.py:127-141: stride 2, no padding).  w fp32 [C][3][3] (torch's [C,1,3,3]), bias fp32 [C] or NULL.
 * stats (optional): per-128-pixel-tile column sums / sums of squares of the rounded outputs, the layout mi_gbn_finalize takes. */
size_t mi_gdwconv_stats_elems(int B, int Ho, int Wo, int C);
int mi_gdwconv(const void* x, long ldx, const float* w, const float* bias, void* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad,
               float* stats, void* stream);
int mi_gdwconv_dgrad(const void* dy, long ldy, const float* w, void* dx, long lddx, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, void* stream);
size_t mi_gdwconv_wgrad_workspace(int B, int Ho, int Wo, int C);
int mi_gdwconv_wgrad(const void* dy, long ldy, const void* x, long ldx, float* dw, float* dbias, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* Criss-cross attention core (contextagg/ccnet.py:56-127) on the projected q, k [B][H][W][Cq] and v [B][H][W][C]: for every pixel the affinities
 * with its column (own position masked with -inf) and its row, ONE softmax over the H + W candidates (att, fp32 [B][H][W][H+W], kept for the
 * backward pass) and agg = sum_j att_j v_j.  The module's output is gamma * agg + x (mi_gbn_apply with scale = gamma).  H + W <= 512.
 * backward: dagg -> dq, dk, dv; de_ws: float[B*H*W*(H+W)] scratch. */
int mi_gcca_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, float* att, void* agg, long ldo, int B, int H, int W, int Cq, int C,
                void* stream);
int mi_gcca_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const float* att, const void* dagg, long lddagg, float* de_ws, void* dq,
                long lddq, void* dk, long lddk, void* dv, long lddv, int B, int H, int W, int Cq, int C, void* stream);
/* Sigmoid gate of the local attention module (GALDNet.py:150-157).  dout == NULL: o1 = x + x * sigmoid(g).  Otherwise the backward:
 * o1 = d loss / d x = dout * (1 + s), o2 = d loss / d g = dout * x * s * (1 - s). */
int mi_ggate(const void* x, long ldx, const void* g, long ldg, const void* dout, long lddo, void* o1, long ld1, void* o2, long ld2, long M, int C, void* stream);

/* mi_gconv (forward, bf16 out, tile statistics) with BatchNorm2d's finalize done by the launch's last workgroup instead of a second launch
 * (mi_gbn_finalize's arithmetic and results: mean, invstd, scale, shift into fin_out[4][N], running statistics updated), for convs of at most
 * mi_gconv_bn_inlaunch_max_pixels() output pixels - the small maps of the deep stages, where the extra launch cost as much as the conv.
 * tickets: one zeroed 32-bit word per 32 output channels, owned by the caller, left zero by the launch (reusable on the same stream). */
int mi_gconv_bn_inlaunch_max_pixels(void);
int mi_gconv_bn(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, const float* bias, float* stats, unsigned* tickets, const float* gamma,
                const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* fin_out, void* stream);

/* ---- the general family in the reference's precision (csrc/gf32.hip): evaluation forward of PraNet / GALD in fp32 ------------------------
 * Replaces the eval()-mode forward of core/testers/pranet_tester.py:36 (`self.model(x)`) and core/testers/gald_tester.py:56-57
 * (`self.encoder(x)`, `self.decoder(x, ...)`), whose masks the testers threshold: fp32 NHWC views (pointer to channel 0 + floats per pixel
 * row), weights read from the fp32 OIHW masters, fp32 accumulation on v_mfma_f32_16x16x4_f32 (exact fp32 products: BASELINE's 1e-3 / identical
 * argmax bar).  Forward only.
 * mi_gconv_f32: out = act(((conv(a, w) + bias) * scale + shift) + add): nn.Conv2d (any taps, per-axis stride / padding / dilation) + its bias,
 * BatchNorm2d in eval() as the affine of mi_gbn_fold (scale / shift NULL: none), the residual view (NULL: none), act 0 none | 1 ReLU | 2 ReLU6 -
 * the order of Res2Net_v1b.py:86-92 / PraNet_Res2Net.py:17-20,57-58 / hardnet_68.py:56-80. */
int mi_gconv_f32(const float* a, long lda, const float* w_oihw, const float* bias, const float* scale, const float* shift, const float* add, long ldadd, int act,
                 float* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                 void* stream);
/* mode 0: AvgPool2d(k, stride, pad), count_include_pad=True (Res2Net_v1b.py:40); 1: AvgPool2d(stride, stride, ceil_mode=True,
 * count_include_pad=False) (Res2Net_v1b.py:122-123); 2: MaxPool2d(k, stride, pad) (Res2Net_v1b.py:152, hardnet_68.py:213,233) */
int mi_gpool_f32(const float* x, long ldx, float* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int mode, void* stream);
/* depthwise 3x3 + bias + eval()-BatchNorm affine + activation (GALDNet.py:127-141) */
int mi_gdwconv_f32(const float* x, long ldx, const float* w, const float* bias, const float* scale, const float* shift, int act, float* out, long ldo, int B, int H,
                   int W, int C, int Ho, int Wo, int stride, int pad, void* stream);
/* criss-cross attention core (ccnet.py:56-127), forward: out = sum_j softmax_j(q . k_j) v_j over the pixel's column (own position masked) and row */
int mi_gcca_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* out, long ldo, int B, int H, int W, int Cq, int C, void* stream);
/* pointwise: op 0 out = act(a * scale[c] + shift[c] (+ b)) (gamma * agg + x of ccnet.py:127); 1 out = (1 - sigmoid(b[m])) * a, b one value per pixel
 * (PraNet_Res2Net.py:131-133); 2 out = a + a * sigmoid(b) (GALDNet.py:150-157); 3 out = relu(a * b) (gcpa_gald.py:88-101) */
int mi_gpoint_f32(int op, const float* a, long lda, const float* b, long ldb, const float* scale, const float* shift, int act, float* out, long ldo, long M, int C,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif
