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
 * bias / add (fp32 [C] / fp32 [N,H,W,C]) may be NULL.  fp32 accumulation. */
int cnx_dwconv7x7_nhwc(const void* x, int x_dtype, const float* w49c, const float* bias,
                       const float* add, void* out, int out_dtype,
                       int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, void* stream);

/* Filter / bias gradient of the same convolution:
 *   dw49c[(kh*7+kw)*C + c] = sum_{n,h,w} dy[n,h,w,c] * x[n,h+kh-3,w+kw-3,c];  dbias[c] = sum dy.
 * Deterministic two-stage reduction; ws is scratch of cnx_dwconv7x7_wgrad_ws_floats(C) floats. */
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

/* Fused ConvNeXt MLP (models/convnext.py:42-49), two chained bf16 MFMA GEMMs with the 4C-wide hidden
 * activation kept in registers:
 *     out[m, :] = resid[m, :] + gamma * (GELU(A[m, :] W1^T + b1) W2^T + b2)
 * A [M, C] bf16; W1 [4C, C] bf16 (nn.Linear layout); W2p [C, 4C] bf16 = W2 with its hidden index
 * permuted inside every group of 32: position t*16 + half*8 + e holds h = (e&3) + 8*(2t + (e>>2)) + 4*half
 * (the order the MFMA accumulators enumerate it); b1 [4C], b2 [C], gamma [C] fp32 (gamma, resid may be
 * NULL); resid / out [M, C] fp32 or bf16; y2_out (nullable) receives the pre-gamma fc2 output in bf16
 * (needed for d(gamma) in the training backward).  Exact-erf GELU (|err| <= 1.5e-7), fp32 accumulate.
 * cnx_mlp_fwd_supported(C) tells which widths have a kernel (96, 192, 384). */
int cnx_mlp_fwd_supported(int32_t C);
int cnx_mlp_fwd(const void* A, const void* W1, const float* b1, const void* W2p, const float* b2,
                const float* gamma, const void* resid, int resid_dtype, void* out, int out_dtype,
                void* y2_out, int64_t M, int32_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONVNEXT_HIP_H_ */
