/*
 * convnext_hip.h — C ABI of the model-side gfx950 kernels in libapgd_hip.so
 * (the ConvNeXt block of /root/reference/models/convnext.py:15-50 and
 * the channels-first LayerNorm(+GELU) of /root/reference/utils_architecture.py:57-81, 127-140).
 *
 * The reference runs these as eager ATen / cuDNN ops; on MI355X the depthwise 7x7 in NHWC
 * bf16 falls to MIOpen's naive solvers (profiles/r01_step0_eager_kernel_stats.md: 63 % of a
 * step), so they are hand-written here.  Same conventions as apgd_hip.h: device pointers,
 * caller-owned memory, hipStream_t as void*, int status, no allocation, no sync.
 *
 * Layouts: activations are channels-last rows, [N,H,W,C] == [M = N*H*W, C] with C contiguous.
 * dtype codes: APGD_F32 (0), APGD_BF16 (1).
 */
#ifndef CONVNEXT_HIP_H_
#define CONVNEXT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Depthwise 7x7, stride 1, zero padding 3 (models/convnext.py:28, 39) on NHWC data.
 *   out[n,h,w,c] = bias[c] + add[n,h,w,c] + sum_{kh,kw} x[n,h+kh-3,w+kw-3,c] * w49c[(kh*7+kw)*C + c]
 * w49c is the filter in tap-major [49][C] fp32 order.  flip != 0 applies the 180-degree
 * rotated filter: with x := d(loss)/d(out) this is the gradient w.r.t. the forward input.
 * bias / add (fp32 [C] / fp32 [N,H,W,C]) may be NULL.  fp32 accumulation.  An all-fp32 call is an exact fp32 stencil;
 * a call with a bf16 operand (x or out) is the autocast convolution: x AND the filter are rounded to bf16 first
 * (packed bf16 dot products, v_dot2c_f32_bf16),, as the reference run under autocast computes its convolutions. */
int cnx_dwconv7x7_nhwc(const void* x, int x_dtype, const float* w49c, const float* bias,
                       const float* add, void* out, int out_dtype,
                       int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, void* stream);

/* Whether calls with a bf16 operand, C % 32 == 0 and W >= 7 take the sliding-window kernels (csrc/dwwin_kernels.hip: forward / input
 * gradient through LDS-DMA, filter gradient) instead of the LDS-ring kernels: 0 = no, anything else = yes (default; APGD_DW_WIN
 * overrides the default at start-up).  Same arithmetic either way (bf16 operands, fp32 accumulation; the summation order differs).
 * Returns the previous setting; a negative `policy` only queries.  Process-wide, not synchronised: set it before launching. */
int cnx_dwconv7x7_win_policy(int policy);

/* Process-wide kernel-selection switches that a RUNNING process may flip between launches (bench.py's interleaved A/B leg times the
 * default tree against the round-4 kernel set on one box, in one process; the environment variables named below only set the
 * start-up values).  Returns the previous value; a negative `value` only queries; an unknown `which` returns -1.  Not synchronised:
 * set it between launches, and drop captured hipGraphs that contain the kernels concerned.  No switch changes results beyond the
 * summation order of the kernels it selects.
 *   CNX_SWITCH_BLK2_WIDTHS     widths served by the wavefront-pair forward blk2_fwd_kernel: bit 0 = C 256, bit 1 = C 384, bit 2 = C 192 (APGD_BLK2)
 *   CNX_SWITCH_DW_SHARED_HALO  shared column halo of the 32-channel sliding-window depthwise wavefronts, 0 / 1 (APGD_DW_SH)
 *   CNX_SWITCH_BLK2_BWD_WIDTHS widths served by the wavefront-pair Hpre backward blk2_bwd_kernel (round 6): bit 0 = C 256, bit 1 = C 384, bit 2 = C 192
 *                              (APGD_BLK2B)
 *   CNX_SWITCH_GEMM_NT_TILE    workgroup tile of cnx_gemm_nt: 0 = by the grid-size rule (default); 1 / 2 / 3 force 128 x 192 / 256 x 192 / 256 x 256
 *                              (256 x 192 where N is no multiple of 256) - measurement runs
 *   CNX_SWITCH_TN_PAIR_RING    stage loop of cnx_gemm_tn_pair (bit-identical results): 1 = two 64-row buffers with the stage hand-over one k-step
 *                              early (the next stage's first fragments are read under the last MFMAs; default), 0 = hand-over at the stage
 *                              boundary (the loop of the single contractions), 2 = a ring of four or more 32-row stages (three or more in flight)
 *   CNX_SWITCH_FWD_WAVES8      widths whose single-wavefront-per-tile forward runs eight wavefronts (256 rows) per workgroup on one weight
 *                              stream instead of four: bit 0 = C 128, bit 1 = C 192 (APGD_FWD_W8).  A measured negative of round 6
 *                              (profiles/r06_fused_mlp.md): only in libraries built with -DBLK_FWD_W8_BUILD=1, otherwise -1 */
#define CNX_SWITCH_BLK2_WIDTHS 0
#define CNX_SWITCH_DW_SHARED_HALO 1
#define CNX_SWITCH_FWD_WAVES8 2
#define CNX_SWITCH_BLK2_BWD_WIDTHS 3
#define CNX_SWITCH_GEMM_NT_TILE 4
#define CNX_SWITCH_TN_PAIR_RING 5
int cnx_runtime_switch(int32_t which, int32_t value);

/* Filter / bias gradient of the same convolution:
 *   dw49c[(kh*7+kw)*C + c] = sum_{n,h,w} dy[n,h,w,c] * x[n,h+kh-3,w+kw-3,c];  dbias[c] = sum dy.
 * Deterministic two-stage reduction; ws is scratch of cnx_dwconv7x7_wgrad_ws_floats(C) floats.  With a bf16 dy (the
 * autocast backward) x is taken in bf16 as well (an fp32 x is rounded, as the forward convolution did) and the sums run
 * on packed bf16 dot products with fp32 accumulation; other dtype pairs use fp32 FMAs on the stored operands. */
int64_t cnx_dwconv7x7_wgrad_ws_floats(int32_t C);
int cnx_dwconv7x7_wgrad_nhwc(const void* x, int x_dtype, const void* dy, int dy_dtype,
                             float* dw49c, float* dbias, float* ws,
                             int64_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* LayerNorm over the last dim of [M, C] rows, eps inside the sqrt, biased variance
 * (F.layer_norm / utils_architecture.py:76-81), optional exact-erf GELU epilogue (the ConvStem
 * pair LN_cf -> GELU).  mean / rstd (fp32 [M]) are saved for the backward (may be NULL). */
int cnx_layernorm_fwd(const void* x, int x_dtype, const float* weight, const float* bias, float eps,
                      void* y, int y_dtype, float* mean, float* rstd,
                      int64_t M, int32_t C, int32_t gelu, void* stream);

/* Backward of the above.  dx = d(loss)/dx; if dweight != NULL also the parameter gradients
 * (two-stage deterministic reduction; ws = cnx_layernorm_bwd_ws_floats(C) floats). */
int64_t cnx_layernorm_bwd_ws_floats(int32_t C);
int cnx_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype,
                      const float* weight, const float* bias, const float* mean, const float* rstd,
                      void* dx, int dx_dtype, float* dweight, float* dbias, float* ws,
                      int64_t M, int32_t C, int32_t gelu, void* stream);
/* The same with the gradient that reached x along a skip connection summed in:  dx = add + d(loss)/dx  (add: fp32 [M, C],
 * dx fp32).  In the transformer blocks of the ViT models (timm Block.forward: x + ls(attn(norm(x)))) x feeds both the
 * LayerNorm and the residual sum, and autograd would add the two gradients in a separate pass over the residual stream. */
int cnx_layernorm_bwd_add(const void* dy, int dy_dtype, const void* x, int x_dtype,
                          const float* weight, const float* bias, const float* mean, const float* rstd,
                          const float* add, void* dx, int dx_dtype, float* dweight, float* dbias, float* ws,
                          int64_t M, int32_t C, int32_t gelu, void* stream);

/* The downsample layers of the stages (models/convnext.py:76-83: LayerNorm over C, then Conv2d(C, C', kernel 2, stride 2)) as
 * LayerNorm + GEMM: cnx_layernorm_fwd_patch2 writes LN(x) of an [N, H, W, C] tensor in 2x2-patch form
 *     y [N*H/2*W/2, 4C],  y[(n, h/2, w/2)][((h&1)*2 + (w&1))*C + c] = LN(x)[n, h, w, c]
 * so that the convolution is  y @ Wp^T + b  with Wp [C', 4C] = weight.permute(0, 2, 3, 1) (a library GEMM with the bias in its
 * epilogue); cnx_layernorm_bwd_patch2 reads the GEMM's input gradient dy in the same form.  mean / rstd [N*H*W] in row order.
 * H, W even; C in {48, 96, 192, 384, 768} (the wide kernels); otherwise APGD_ERR_ARG. */
int cnx_layernorm_fwd_patch2(const void* x, int x_dtype, const float* weight, const float* bias, float eps, void* y,
                             int y_dtype, float* mean, float* rstd, int64_t N, int32_t H, int32_t W, int32_t C, void* stream);
int cnx_layernorm_bwd_patch2(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* mean,
                             const float* rstd, void* dx, int dx_dtype, float* dweight, float* dbias, float* ws, int64_t N,
                             int32_t H, int32_t W, int32_t C, void* stream);

/* First ConvStem convolution: Conv2d(3, P, kernel 3, stride 2, padding 1) (utils_architecture.py:127, 180, 205) on the
 * attack state itself.  x [N, 3, H, W] fp32 NCHW contiguous -> out [N, OH, OW, P] bf16 rows (OH = ceil(H/2)); autocast
 * numerics (x and the filter rounded to bf16, fp32 accumulation; MFMA).  wq = cnx_stem_conv_pack(weight [P,3,3,3]) is the
 * filter as MFMA operand fragments for both directions, cnx_stem_conv_packed_bytes(P) bytes.  cnx_stem_conv_dgrad: dy [N, H/2, W/2, P] bf16 -> dx [N, 3, H, W] fp32 NCHW (H, W
 * even) = the gradient the APGD update consumes.  P in {48, 64, 96}. */
int cnx_stem_conv_supported(int32_t P);
int64_t cnx_stem_conv_packed_bytes(int32_t P);
int cnx_stem_conv_pack(const void* w, int w_dtype, void* wq, int32_t P, void* stream);
int cnx_stem_conv_fwd(const float* x, const void* wq, const float* bias, void* out,
                      int64_t N, int32_t H, int32_t W, int32_t P, void* stream);
/* Filter / bias gradient of the same convolution (utils_architecture.py:205-211 backward; replaces the library's
 * convolution_backward = MIOpen, whose per-process solver search ran seconds inside every rank's warm-up):
 *   dw[p][ci][kh][kw] = sum_{n,oh,ow} dy[n,oh,ow,p] * bf16(x[n,ci,2oh-1+kh,2ow-1+kw]);  dbias[p] = sum dy      (fp32 results)
 * x: the fp32 NCHW image batch, dy: bf16 [N, H/2, W/2, P] rows; bf16 operands, fp32 accumulation on the matrix pipe (the
 * contraction runs over the positions: dy rows reach the MFMA through LDS transpose reads, see csrc/wgrad_kernels.hip).
 * Deterministic: per-workgroup partial results in ws (cnx_stem_conv_wgrad_ws_floats(P) floats) summed in a fixed order.
 * dbias may be NULL.  Same size limits as cnx_stem_conv_fwd. */
int64_t cnx_stem_conv_wgrad_ws_floats(int32_t P);
int cnx_stem_conv_wgrad(const float* x, const void* dy, float* dw, float* dbias, float* ws,
                        int64_t N, int32_t H, int32_t W, int32_t P, void* stream);

/* The same convolution with the ConvStem's LayerNorm(channels_first) + GELU (utils_architecture.py:76-81, 128-129) applied to
 * the output tile while it is still on chip: act [N, H/2, W/2, P] bf16 = GELU(LN(conv(x) rounded to bf16)); y (nullable: a
 * gradient-free forward does not need it) = the convolution output, mean / rstd (nullable pair) [N*H/2*W/2] the LayerNorm
 * statistics - what cnx_layernorm_bwd(gelu = 1) takes in the backward. */
int cnx_stem_conv_ln_gelu_fwd(const float* x, const void* wq, const float* bias, const float* ln_w, const float* ln_b, float eps,
                              void* y, void* act, float* mean, float* rstd, int64_t N, int32_t H, int32_t W, int32_t P,
                              void* stream);
int cnx_stem_conv_dgrad(const void* dy, const void* wq, float* dx,
                        int64_t N, int32_t H, int32_t W, int32_t P, void* stream);
/* The same input gradient reduced to its SIGN {-1, 0, +1} (int8, [N, 3, H, W]): the Linf APGD update
 * (autopgd_train_clean.py:221: step_size * sign(grad)) reads nothing else of it, so the attack asks for this
 * form and apgd_linf_step_f32(grad_dtype = APGD_I8) / apgd_track_rows(grad_elt = 1) move a quarter of the gradient bytes.
 * sign is taken of the same fp32 accumulator cnx_stem_conv_dgrad stores: identical decisions. */
int cnx_stem_conv_dgrad_sign(const void* dy, const void* wq, int8_t* sign_out,
                             int64_t N, int32_t H, int32_t W, int32_t P, void* stream);
/* The same signs in the BLOCKED order apgd_linf_step_f32(grad_dtype = APGD_I8_BLK) reads (apgd_hip.h): inside every group of
 * 1024 elements of a sample, element g*1024 + (u*64 + l)*4 + j (u < 4, l < 64, j < 4) is stored at byte g*1024 + l*16 + u*4 + j,
 * so that a lane of the update kernel finds the signs of its four float4 chunks in ONE 16-byte load.  Needs 3*H*W % 1024 == 0. */
int cnx_stem_conv_dgrad_sign_blk(const void* dy, const void* wq, int8_t* sign_out,
                                 int64_t N, int32_t H, int32_t W, int32_t P, void* stream);

/* Element-wise tails of the MLP for widths that run their GEMMs in the library (models/convnext.py:44-49 and their
 * backward), one pass each, per-channel parameter gradients accumulated on the way (deterministic two-stage sums;
 * ws = cnx_colsum_ws_floats(n_cols) floats of scratch):
 *   cnx_scale_residual      out = x + gamma * y                       x, out fp32 or bf16 [M, C]; y bf16; gamma nullable
 *   cnx_scale_residual_bwd  dos = bf16(g * gamma) (dos NULL: sums only);  dgamma[c] = sum_m g*y, db2[c] = sum_m dos
 *                           (both NULL or both set)
 *   cnx_gelu_bwd_colsum     dhpre = bf16(dh * GELU'(hpre)) on bf16 [M, N];  db1[n] = sum_m dhpre (nullable) */
int64_t cnx_colsum_ws_floats(int32_t n_cols);

/* out[j] = sum_s parts[s][j]: fp32 sum (fixed order) of the S bf16 partial products [S, L] of a split-K batched GEMM - the
 * weight gradients dW = X^T dY of models/convnext.py:42-46 contract over N*H*W rows (up to 802 816), which the product runs as
 * S batches of a library GEMM.  L % 8 == 0. */
int cnx_sum_parts_bf16(const void* parts, float* out, int64_t S, int64_t L, void* stream);
int cnx_scale_residual(const void* x, int x_dtype, const void* y, const float* gamma, void* out, int out_dtype,
                       int64_t M, int32_t C, void* stream);
int cnx_scale_residual_bwd(const void* g, int g_dtype, const void* y, const float* gamma, void* dos,
                           float* dgamma, float* db2, float* ws, int64_t M, int32_t C, void* stream);
int cnx_gelu_bwd_colsum(const void* dh, const void* hpre, void* dhpre, float* db1, float* ws,
                        int64_t M, int32_t N, void* stream);
/* d(gamma) of a ConvNeXt block (models/convnext.py:47: x = self.gamma * x) from the weight gradients of its second linear layer,
 * without a pass over the [M, C] tensors:  with y2 = H W2^T + b2 and dO = bf16(g gamma),
 *     dgamma[c] = sum_m g[m,c] y2[m,c] = (sum_j bf16(W2[c,j]) dW2[c,j] + b2[c] db2[c]) / gamma[c]
 * w2, dw2 fp32 [C, Hd] (dW2 = dO^T H), b2, db2 (= sum_m dO; both nullable together), gamma fp32 [C].  g (fp32 / bf16 [M, C]) and y2
 * (bf16 [M, C]; or, y2 == NULL, h_tiles = H in the CNX_TN_ACC tiles of the fused kernels, from which y2 is recomputed: M, Hd multiples
 * of 32) are read only for channels with |gamma| < 1e-30 (dO = bf16(g gamma) has flushed: the direct sum; the layer-scale init 1e-6 and
 * anything a model can hold is served by the identity).  No cancellation guard is needed: the two terms add up to sum_m dO y2 exactly,
 * so the identity's error is the bf16 rounding of dO, the same size as the reference's own rounding of y2.  Replaces cnx_scale_residual_bwd's sums-only mode in the
 * training pass of the fused blocks, whose forward then has no pre-gamma output to keep. */
int cnx_block_dgamma(const float* w2, const float* dw2, const float* b2, const float* db2, const float* gamma,
                     const void* g, int g_dtype, const void* y2, const void* h_tiles, float* dgamma,
                     int64_t M, int32_t C, int32_t Hd, void* stream);
/* The LayerNorm parameter gradients of a block (models/convnext.py:40-41 backward) from the first linear layer's weight gradients:
 *     dlb[c] = sum_m da[m,c] = sum_j bf16(W1[j,c]) db1[j],     dlw[c] = (sum_j bf16(W1[j,c]) dW1[j,c] - ln_b[c] dlb[c]) / ln_w[c]
 * w1, dw1 fp32 [Hd, C] (dW1 = dHpre^T LN(u)), db1 fp32 [Hd].  The dlw identity recovers xh from LN(u) = bf16(xh ln_w + ln_b): its error is
 * ~2^-9 max(1, |ln_b| / |ln_w|) per channel.  A channel with |ln_b| > 4 |ln_w| (or ln_w == 0, or NaN parameters) is ill-conditioned for
 * it and gets the DIRECT sum dlw[c] = sum_m da[m,c] xh[m,c] instead, from da (bf16 [M, C], gradient w.r.t. LN(u); or, da == NULL,
 * dhpre_tiles = dHpre in CNX_TN_ACC tiles, from which da is recomputed for those channels), u (bf16 [M, C]), mean, rstd [M] - which are
 * read for such channels only.  ws: cnx_block_dln_ws_floats(C) floats for the direct sums (a 256-workgroup pass that leaves at once
 * when no channel is ill-conditioned, fixed-order partials: deterministic); ws == NULL: the direct sums run serially inside the small
 * kernel (correct, slow).  The LayerNorm backward then needs no partial sums of its own - and rides in the epilogue of the block's
 * backward kernel: cnx_block_mlp_bwd_train_hpre_ln / cnx_block_mlp_bwd_acc_ln = cnx_block_mlp_bwd_train_hpre / cnx_block_mlp_bwd_acc
 * with du = d(loss)/du (the gradient w.r.t. the depthwise-conv output) written where those write da. */
int64_t cnx_block_dln_ws_floats(int32_t C);
int cnx_block_dln(const float* w1, const float* dw1, const float* db1, const float* ln_w, const float* ln_b,
                  const void* da, const void* dhpre_tiles, const void* u, const float* mean, const float* rstd,
                  float* dlw, float* dlb, float* ws, int64_t M, int32_t C, int32_t Hd, void* stream);
int cnx_block_mlp_bwd_train_hpre_ln(const void* u, const float* ln_w, const float* mean, const float* rstd,
                                    const void* g, int g_dtype, const float* gamma, const void* Wb, const void* hpre_ws,
                                    void* du, void* do_rows, void* dhpre_ws, int64_t M, int32_t C, void* stream);
int cnx_block_mlp_bwd_acc_ln(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                             const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* du,
                             void* a_rows, void* do_rows, void* h_ws, void* dhpre_ws, int64_t M, int32_t C, void* stream);
/* y = GELU(x) (exact-erf form, nn.GELU() of models/convnext.py:43 and of the timm Mlp) on n bf16 elements, n % 8 == 0. */
int cnx_gelu_fwd(const void* x, void* y, int64_t n, void* stream);

/* Fused block tail (models/convnext.py:40-49): LayerNorm -> fc1 -> GELU -> fc2 -> gamma -> +residual, ONE kernel:
 *     out[m, :] = resid[m, :] + gamma * (GELU(LN(u[m, :]) W1^T + b1) W2^T + b2)
 * u [M, C] bf16 = depthwise-conv output; LN (eps inside the sqrt, fp32 statistics, two-pass) is applied when
 * ln_w != NULL (mean / rstd [M] fp32 are then written if non-NULL), otherwise u is taken as already normalised.
 * Wf = cnx_mlp_pack_weights(W1 [4C, C], W2 [C, 4C]) : bf16 weights in MFMA-fragment order, cnx_mlp_packed_elems(C)
 * elements (8 C^2, or 8 C^2 + 64 C where the kernel's hidden loop is software-pipelined and wants slice t = [W1 block t |
 * W2 block t-1]: always size the buffer with cnx_mlp_packed_elems; re-pack after every optimizer step).  b1 [4C], b2 [C], gamma [C] fp32 (gamma, resid may be NULL);
 * resid / out fp32 or bf16; y2_out (nullable) = pre-gamma fc2 output in bf16 (for d(gamma)).
 * GELU is the exact-erf form evaluated with |error| <= 1.2e-6; bf16 MFMA, fp32 accumulate. */
int cnx_block_mlp_supported(int32_t C);
int64_t cnx_mlp_packed_elems(int32_t C);
int cnx_mlp_pack_weights(const void* W1, const void* W2, int w_dtype, void* Wf, int32_t C, void* stream);
int cnx_block_mlp_fwd(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                      const void* Wf, const float* b1, const float* b2, const float* gamma,
                      const void* resid, int resid_dtype, void* out, int out_dtype, void* y2_out,
                      int64_t M, int32_t C, void* stream);

/* Backward of the block tail w.r.t. a = LN(u) (the input-gradient chain of models/convnext.py:41-49), ONE kernel:
 *     dO = g * gamma;  dH = dO W2;  Hpre = a W1^T + b1 (recomputed);  dHpre = dH * GELU'(Hpre);  da = dHpre W1
 * u, ln_w, ln_b, mean, rstd as in / from the forward; g [M, C] fp32 or bf16 = gradient w.r.t. the block output;
 * Wb = cnx_mlp_pack_weights_bwd(W1, W2), cnx_mlp_packed_bwd_elems(C) bf16 elements; da [M, C] bf16 (feed it to
 * cnx_layernorm_bwd).  Training backward: pass all four emit pointers to also get the operands of the weight
 * gradients: a_out, do_out [M, C] bf16 and ht_out = GELU(Hpre)^T, dhpt_out = dHpre^T as [4C, M] bf16
 *     dW1 = dHpre^T a,  db1 = rowsum(dHpre^T),  dW2 = (H^T dO)^T,  db2 = colsum(dO).
 * a_stride (elements, 0 = C, multiple of 8) is the row stride of a_out: with a ones column appended by the caller
 * (a_out [M, C+8]) the d(b1) sum rides along in the dW1 GEMM.
 * cnx_block_mlp_bwd_supported(C): widths with a kernel (96, 128, 192, 256). */
int cnx_block_mlp_bwd_supported(int32_t C);
int64_t cnx_mlp_packed_bwd_elems(int32_t C);
int cnx_mlp_pack_weights_bwd(const void* W1, const void* W2, int w_dtype, void* Wb, int32_t C, void* stream);
int cnx_block_mlp_bwd(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                      const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* da,
                      void* a_out, int64_t a_stride, void* do_out, void* ht_out, void* dhpt_out,
                      int64_t M, int32_t C, void* stream);
/* Input-gradient-only variant (the attack's backward, models/convnext.py:41-49 including the LayerNorm): the same kernel
 * with the LayerNorm backward in its epilogue,
 *     du = rstd * (t - mean_c(t) - xh * mean_c(t * xh)),   t = ln_w * da,  xh = (u - mean) * rstd,
 * so the [M, C] gradient is written once, as du [M, C] bf16 = gradient w.r.t. the depthwise-conv output (feed it to
 * cnx_dwconv7x7_nhwc(flip = 1)); no parameter gradients. */
int cnx_block_mlp_bwd_input(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                            const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* du,
                            int64_t M, int32_t C, void* stream);

/* The same pair with Hpre handed over through HBM instead of recomputed: the forward ALSO writes
 * Hpre = LN(u) W1^T + b1 (what models/convnext.py:42 hands to GELU) as bf16 into a caller-allocated workspace of
 * cnx_block_mlp_hpre_elems(M, C) elements, in the accumulator order of the kernels (opaque to the caller), and
 * cnx_block_mlp_bwd_input_hpre reads it back: two GEMMs per hidden slice instead of three and no LN(u) operand fragments
 * in registers (C = 384, whose recomputing kernel does not fit one wavefront per SIMD, exists only in this form; C = 192
 * runs two wavefronts per SIMD in this form, one in the recomputing one).  Worth its 2 x 8 C bytes per row of extra HBM
 * traffic where the block is not HBM-bound: measured 15 - 19 % less time for the pair at C = 128, 192, 256.  Same
 * arguments otherwise (no y2_out: input-gradient passes only; mean / rstd are required).
 * cnx_block_mlp_hpre_supported(C): 128, 192, 256, 384. */
int cnx_block_mlp_hpre_supported(int32_t C);
int64_t cnx_block_mlp_hpre_elems(int64_t M, int32_t C);
int cnx_block_mlp_fwd_hpre(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                           const void* Wf, const float* b1, const float* b2, const float* gamma,
                           const void* resid, int resid_dtype, void* out, int out_dtype, void* hpre_ws,
                           int64_t M, int32_t C, void* stream);
int cnx_block_mlp_bwd_input_hpre(const void* u, const float* ln_w, const float* mean, const float* rstd,
                                 const void* g, int g_dtype, const float* gamma, const void* Wb, const void* hpre_ws,
                                 void* du, int64_t M, int32_t C, void* stream);

/* The TRAINING pass on the same kernel pair (round 5; C = 128 ... 384): forward and backward hand the weight-gradient contractions
 * their operands in the layouts cnx_gemm_tn_ex reads, so that no GEMM of the block is a library call and nothing is transposed:
 *   cnx_block_mlp_fwd_train      = cnx_block_mlp_fwd_hpre + H = GELU(Hpre) into a second workspace of the same size and tiles
 *                                  (CNX_TN_ACC), the LN(u) rows a_rows [M, C] bf16, and (optionally) y2_out like cnx_block_mlp_fwd;
 *   cnx_block_mlp_bwd_train_hpre   from g, gamma, the packed backward weights and the Hpre workspace: da [M, C] bf16 = gradient w.r.t.
 *                                  LN(u) (the LayerNorm backward with its parameter gradients is cnx_layernorm_bwd), do_rows [M, C]
 *                                  bf16 = g * gamma, dhpre_ws = dHpre as CNX_TN_ACC tiles (cnx_block_mlp_hpre_elems elements).
 * Then dW2, db2 = cnx_gemm_tn_ex(A = do_rows ROWS, B = h_ws ACC), dW1, db1 = cnx_gemm_tn_ex(A = dhpre_ws ACC, B = a_rows ROWS).
 *   cnx_block_mlp_bwd_acc        the recomputing training backward (cnx_block_mlp_bwd; C = 96 ... 256) with H and dHpre written as
 *                                  CNX_TN_ACC tiles instead of transposed [4C, M] matrices (M a multiple of 32; workspaces of
 *                                  M * 4C elements each). */
int cnx_block_mlp_fwd_train(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                            const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                            int resid_dtype, void* out, int out_dtype, void* y2_out, void* hpre_ws, void* h_ws, void* a_rows,
                            int64_t M, int32_t C, void* stream);
int cnx_block_mlp_bwd_train_hpre(const void* g, int g_dtype, const float* gamma, const void* Wb, const void* hpre_ws, void* da,
                                 void* do_rows, void* dhpre_ws, int64_t M, int32_t C, void* stream);
int cnx_block_mlp_bwd_acc(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                          const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* da,
                          void* a_rows, void* do_rows, void* h_ws, void* dhpre_ws, int64_t M, int32_t C, void* stream);

/* Second ConvStem convolution (utils_architecture.py:207-209 `ConvBlock1`: 48 -> 96; `ConvBlock3`: 64 -> 96): 3x3, stride 2, padding 1
 * on channels-last bf16 activations, forward and input gradient as implicit GEMMs on MFMA (filter resident in LDS, activations as
 * 16-byte loads straight from the NHWC tensor).  x [N, H, W, CI] bf16 -> out [N, H/2, W/2, CO] bf16 (+ bias [CO] fp32, nullable);
 * dgrad: dy [N, H/2, W/2, CO] bf16 -> dx [N, H, W, CI] bf16.  packed = cnx_conv3x3s2_pack(w [CO, CI, 3, 3] fp32 or bf16):
 * cnx_conv3x3s2_packed_elems(CI, CO) bf16 elements.  Results are reproducible from run to run (the library's backward-data
 * kernel for this layer is not).  cnx_conv3x3s2_supported: CO == 96, CI in {48, 64}, H % 8 == 0, W % 16 == 0. */
int cnx_conv3x3s2_supported(int32_t CI, int32_t CO, int32_t H, int32_t W);
int64_t cnx_conv3x3s2_packed_elems(int32_t CI, int32_t CO);
int cnx_conv3x3s2_pack(const void* w, int w_dtype, void* packed, int32_t CI, int32_t CO, void* stream);
int cnx_conv3x3s2_fwd(const void* x, const void* packed, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                      int32_t CI, int32_t CO, void* stream);
int cnx_conv3x3s2_dgrad(const void* dy, const void* packed, void* dx, int64_t N, int32_t H, int32_t W, int32_t CI, int32_t CO,
                        void* stream);

/* Filter / bias gradient of the same convolution (utils_architecture.py:205-211 backward; the library's convolution_backward = MIOpen
 * before round 5):  dw[co][kh][kw][ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * x[n,2oh-1+kh,2ow-1+kw,ci]   (fp32, the weight's
 * channels-last order: a [CO, CI, 3, 3] tensor with strides (9 CI, 1, 3 CI, CI));  dbias[co] = sum dy  (dbias may be NULL).
 * x: bf16 [N, H, W, CI] rows, dy: bf16 [N, H/2, W/2, CO] rows.  Implicit GEMM over the positions with both operands read out of LDS
 * by transpose reads (csrc/wgrad_kernels.hip); deterministic (per-workgroup partials in ws, cnx_conv3x3s2_wgrad_ws_floats floats,
 * summed in a fixed order).  Shapes: cnx_conv3x3s2_wgrad_supported (CO = 96, CI in {48, 64}, even H, W up to the LDS budget). */
int cnx_conv3x3s2_wgrad_supported(int32_t CI, int32_t CO, int32_t H, int32_t W);
int64_t cnx_conv3x3s2_wgrad_ws_floats(int32_t CI, int32_t CO);
int cnx_conv3x3s2_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws,
                        int64_t N, int32_t H, int32_t W, int32_t CI, int32_t CO, void* stream);

/* Fused multi-head softmax attention of the ViT family (timm 0.8 `Attention.forward`, reached through the models of
 * /root/reference/utils_architecture.py:272-301):
 *     q, k, v = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4);   out = softmax(q k^T * scale) v
 * qkv [B, N, 3*H*d] bf16 (the packed projection, token-major), out [B, N, H*d] bf16, lse (nullable) [B, H, N] fp32 =
 * log-sum-exp of the scaled scores per query (saved for the backward).  One workgroup per (batch, head); K and V of
 * the head stay in LDS, the N x N scores stay in MFMA accumulators (bf16 MFMA, fp32 softmax).
 * cnx_attention_supported(N, d): d == 64 and N <= 416 (197 tokens @224, 401 @320). */
int cnx_attention_supported(int32_t N, int32_t head_dim);
int cnx_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int32_t N, int32_t H, int32_t head_dim,
                      float scale, void* stream);

/* Backward of the same op: dqkv [B, N, 3*H*d] bf16 from qkv, out, dout ([B, N, H*d] bf16) and the saved lse.  Two
 * kernels stream 32x32 blocks of P = exp(S*scale - lse) through the MFMA accumulators (no N x N tensor in memory):
 * dQ (+ D_q = rowsum(dO*O), written to dvec [B, H, N] fp32 scratch), then dK and dV.  N <= 416, d == 64
 * (N <= 224: every operand image in LDS; above, the dK / dV kernel takes its row operands from memory). */
int cnx_attention_bwd_supported(int32_t N, int32_t head_dim);
int cnx_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dvec,
                      int64_t B, int32_t N, int32_t H, int32_t head_dim, float scale, void* stream);

/* Contraction over the ROW index of two row-major operands (round 5; csrc/wgrad_kernels.hip):
 *     D[N1][N2] = A^T B,   A: bf16 [M, N1] (row stride lda elements), B: bf16 [M, N2] (ldb), D: fp32 [N1, N2] contiguous
 * - the weight gradients of the pointwise convolutions and linears (models/convnext.py:42-46 backward: dW1 = dHpre^T LN(u),
 * dW2 = dO^T H), which were split-K batched library GEMMs (hipBLASLt) plus a partial-sum pass.  Both operands go HBM -> LDS as
 * they lie in memory (LDS-DMA) and reach the MFMA through transpose reads (ds_read_b64_tr_b16); M is split over workgroups, the
 * fp32 partial results (ws: cnx_gemm_tn_ws_floats(M, N1, N2) floats) are added in a fixed order: deterministic.
 * Shapes: M % 64 == 0; (N1, N2) multiples of one of the tiles 384x192, 192x384, 256x128, 128x256, 384x96, 96x384, 256x64, 64x256
 * (cnx_gemm_tn_supported); lda, ldb multiples of 8; 16-byte aligned pointers; operands below 4 GB. */
int cnx_gemm_tn_supported(int64_t M, int32_t N1, int32_t N2);
int64_t cnx_gemm_tn_ws_floats(int64_t M, int32_t N1, int32_t N2);
int cnx_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* D, float* ws,
                int64_t M, int32_t N1, int32_t N2, void* stream);
/* The same contraction with a per-operand layout:
 *   CNX_TN_ROWS  row-major [M][N] with a leading dimension;
 *   CNX_TN_ACC   tiles [M / 32][N / 32] of 2 KiB in the accumulator order of the fused block kernels (element (m, n) of a tile at byte
 *                64 m + 32 ((n / 4) % 2) + 8 (n / 8) + 2 (n % 4)): what cnx_block_mlp_fwd_hpre / the emitting backward kernels write for
 *                the hidden activation H, its pre-activation and dHpre - two 16-byte stores per lane straight from the accumulators,
 *                no transposition pass; the leading dimension is ignored (M a multiple of 32, N = the operand's width).
 * At most one operand is CNX_TN_ACC.  With an ACC operand the call also returns colsum_a[N1] = sum_m A[m][.] (required then): the bias
 * gradient that belongs to the weight gradient (db1 = sum dHpre with dW1 = dHpre^T a; db2 = sum dO with dW2 = dO^T H). */
#define CNX_TN_ROWS 0
#define CNX_TN_ACC 1
int cnx_gemm_tn_ex(const void* A, int64_t lda, int32_t a_layout, const void* B, int64_t ldb, int32_t b_layout,
                   float* D, float* colsum_a, float* ws, int64_t M, int32_t N1, int32_t N2, void* stream);
/* BOTH weight gradients of one block in one launch (round 6): two contractions over the same M rows,
 *     D0 [N1][N2] = A0^T B0,  colsum_a0 [N1] = sum_m A0[m][.]      A0 CNX_TN_ACC tiles [M, N1], B0 rows [M, N2] (row stride ldb0)
 *     D1 [N2][N1] = B1^T A1,  colsum_b1 [N2] = sum_m B1[m][.]      A1 CNX_TN_ACC tiles [M, N1], B1 rows [M, N2] (row stride ldb1)
 * i.e. dW1 = dHpre^T LN(u) with d(b1), and dW2 = dO^T H with d(b2) (A0 = dhpre_ws, B0 = a_rows, A1 = h_ws, B1 = do_rows; N1 = 4C, N2 = C;
 * models/convnext.py:42-46 backward).  A split of M is shared by the tiles of both problems, so the chip is filled with half the splits
 * two cnx_gemm_tn_ex launches need: half the fp32 partial results written and summed (fixed order: deterministic), one launch + one sum
 * instead of two + two.  D1 is accumulated transposed (tile operand on the A side for both) and transposed back by the sum.
 * Shapes: M % 64 == 0 and (N1, N2) multiples of one of the tiles 384x192, 256x128, 384x96, 256x64 (cnx_gemm_tn_pair_supported);
 * ws: cnx_gemm_tn_pair_ws_floats floats; 16-byte aligned pointers; ldb0, ldb1 multiples of 8; operands below 4 GB. */
int cnx_gemm_tn_pair_supported(int64_t M, int32_t N1, int32_t N2);
int64_t cnx_gemm_tn_pair_ws_floats(int64_t M, int32_t N1, int32_t N2);
int cnx_gemm_tn_pair(const void* A0, const void* B0, int64_t ldb0, const void* A1, const void* B1, int64_t ldb1, float* D0,
                     float* colsum_a0, float* D1, float* colsum_b1, float* ws, int64_t M, int32_t N1, int32_t N2, void* stream);

/* bf16 GEMM with a fused epilogue (csrc/gemm_kernels.hip) for the pointwise convolutions / linears without a fused-block kernel:
 * models/convnext.py:42-46 (pwconv1 / act / pwconv2 / gamma / residual) at the widths and passes the fused kernels do not cover,
 * the stage downsample convolutions :76-83 in GEMM form, and the qkv / proj / fc1 / fc2 linears of the transformer blocks
 * (utils_architecture.py:271-301; timm Block).  These ran as hipBLASLt GEMMs between one-pass kernels.
 *     acc[m, n] = sum_k A[m, k] * B[n, k]          A [M, K] bf16 (row stride lda), B [N, K] bf16 (row stride ldb): both K-contiguous,
 *                                                  i.e. B is an nn.Linear weight as stored (or a transposed copy for input gradients)
 *   epilogue 0 (bias)        D = bf16(acc + bias)
 *   epilogue 1 (bias, GELU)  z = bf16(acc + bias);  D = bf16(GELU(z));  z_out (nullable, row stride ldz) = z
 *   epilogue 2 (scale, res.) y = bf16(acc + bias);  D = R + gamma * y  (D, R fp32 or bf16; gamma NULL = 1; R NULL = 0);  z_out (nullable) = y
 *   epilogue 3 (GELU')       D = bf16(bf16(acc) * GELU'(z_in))        (z_in [M, N] bf16, row stride ldz)
 * bias [N] fp32 (rounded to bf16, as the autocast linear adds it) or NULL.  fp32 accumulation; exact-erf GELU / GELU' with the
 * polynomials of the fused block kernels.  Needs K % 64 == 0, N % 4 == 0, 16-byte aligned rows (cnx_gemm_nt_supported). */
int cnx_gemm_nt_supported(int64_t M, int32_t N, int32_t K);
int cnx_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* D, int64_t ldd, int d_dtype, int64_t M, int32_t N,
                int32_t K, int32_t epilogue, const float* bias, const float* gamma, const void* R, int64_t ldr, int r_dtype,
                void* z_out, const void* z_in, int64_t ldz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONVNEXT_HIP_H_ */
