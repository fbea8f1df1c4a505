/*
 * apgd_hip.h — C ABI of libapgd_hip.so, the MI355X (gfx950) kernels behind the
 * `adv.attack=apgd` path of nmndeep/revisiting-at.
 *
 * The reference has no FFI: its boundary is the Python callable
 *     apgd_train(model, x, y, norm, eps, n_iter, ...) -> (x_best, acc, loss_best, x_best_adv)
 * (/root/reference/autopgd_train_clean.py:123-124, 371) injected into WrappedModel
 * (/root/reference/main.py:260-301, 831-844).  The host mirror of that callable lives in
 * revisiting-at_amd/apgd.py and reaches the GPU only through the entry points below.
 * Each entry point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is allocated, freed or
 *    synchronised inside the library; no global state: calls are re-entrant per stream;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *  - attack-state tensors are fp32 "rows": B samples of E contiguous elements each
 *    (any memory format whose outermost dimension is the batch: NCHW or channels_last);
 *  - per-sample vectors have B entries; booleans are uint8 (0/1);
 *  - return value: APGD_OK (0), a negative APGD_ERR_* for bad arguments, or a positive
 *    hipError_t from the launch.  No exceptions cross the boundary.
 *  - arithmetic is IEEE fp32, one rounding per operation in the reference's association
 *    order (built with -ffp-contract=off): results are bit-identical to the reference's
 *    eager ATen sequence for finite inputs.
 */
#ifndef APGD_HIP_H_
#define APGD_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APGD_HIP_VERSION 10800 /* major*10000 + minor*100 + patch */

#define APGD_OK 0
#define APGD_ERR_NULL (-1)    /* required pointer is NULL */
#define APGD_ERR_SIZE (-2)    /* negative / inconsistent size */
#define APGD_ERR_DTYPE (-3)   /* unknown dtype code */
#define APGD_ERR_ARG (-4)     /* other invalid argument */

/* element types for logits / activations */
#define APGD_F32 0
#define APGD_BF16 1
#define APGD_F16 2
#define APGD_I8 3 /* gradient SIGNS {-1, 0, +1} as int8: all the Linf step uses of the gradient (autopgd_train_clean.py:221) */
/* the same signs in BLOCKED order (apgd_linf_step_f32 only; E % 1024 == 0): inside every group of 1024 elements of a sample,
 * element g*1024 + (u*64 + l)*4 + j (u < 4, l < 64, j < 4) is stored at byte g*1024 + l*16 + u*4 + j - a lane of the update
 * kernel reads the signs of its four float4 chunks with one 16-byte load (cnx_stem_conv_dgrad_sign_blk writes this order;
 * apgd_track_rows moves whole rows and does not care) */
#define APGD_I8_BLK 4

/* bits of the per-sample flag byte produced by apgd_state_update */
#define APGD_FLAG_NEW_BEST 1u  /* loss_indiv > loss_best          (autopgd_train_clean.py:321)      */
#define APGD_FLAG_MISCLS 2u    /* ~pred                            (autopgd_train_clean.py:301)      */
#define APGD_FLAG_HALVE 4u     /* fl_oscillation > 0               (autopgd_train_clean.py:340-341)  */

int apgd_hip_version(void);
const char* apgd_hip_strerror(int code);

/* a1 prologue — autopgd_train_clean.py:135, 141-143.
 * x_adv = clamp(x, 0, 1);  x_best = x_adv;  x_best_adv = x_adv   (one pass, three stores).
 * x_best / x_best_adv may be NULL to skip that copy. */
int apgd_init_f32(const float* x, float* x_adv, float* x_best, float* x_best_adv,
                  int64_t n, void* stream);

/* a2 Linf step — autopgd_train_clean.py:213-226, 260.
 *   grad2 = x_adv - x_adv_old
 *   x1    = P(x_adv + step[b]*sign(grad))                      P(t)=clamp(min(max(t,x-eps),x+eps),0,1)
 *   out   = P((x_adv + (x1 - x_adv)*a) + grad2*(1-a))
 * `x_adv_old = x_adv.clone()` (:215) is a buffer rotation on the host: out must not alias an input.
 * grad is fp32, bf16 or APGD_I8 signs (grad_dtype; only its sign is used: with int8 signs the kernel moves 17
 * instead of 20 bytes per element, 13 instead of 16 at i = 0).  out_bf16 (nullable) receives
 * round-to-nearest-even bf16(out) for a bf16 model input.  step_size is [B]. */
int apgd_linf_step_f32(const float* x, const float* x_adv, const float* x_adv_old,
                       const void* grad, int grad_dtype, const float* step_size,
                       float* out, uint16_t* out_bf16,
                       int64_t B, int64_t E, float eps, float a, void* stream);
/* same, with launch-shape knobs for tuning sweeps (<= 0 selects the default):
 * blocks_per_sample = gridDim.x, unroll in {1,2,4} float4 per stream in flight per thread,
 * nontemporal != 0 uses nontemporal loads/stores. */
int apgd_linf_step_f32_ex(const float* x, const float* x_adv, const float* x_adv_old,
                          const void* grad, int grad_dtype, const float* step_size,
                          float* out, uint16_t* out_bf16, int64_t B, int64_t E, float eps, float a,
                          int32_t blocks_per_sample, int32_t unroll, int32_t nontemporal,
                          void* stream);

/* a2 + a4-a6 in one pass (round 5): the Linf step of iteration i+1 together with the row moves iteration i decided on —
 * autopgd_train_clean.py:213-226, 260 and :304, 322-323, 345-346.  Per sample, by flags[b] (apgd_state_update of iteration i):
 *   MISCLS   : x_best_adv <- x_adv
 *   NEW_BEST : x_best <- x_adv, grad_best <- grad
 *   HALVE (without NEW_BEST): the step runs from x_best / grad_best, and x_adv <- x_best (the buffer becomes x_adv_old by the
 *              host's rotation); grad itself is not rewritten - the next backward replaces it (:277-283)
 * then out = the step of apgd_linf_step_f32 on those operands, same arithmetic bit for bit.
 * flags == NULL selects the iteration-0 form (requires x_adv_old == x_adv, a == 1): the prologue's clones x_best = x_best_adv =
 * x_adv (:142-143) and grad_best = grad (:189) are written here, so apgd_init_f32 only has to produce x_adv.
 * grad / grad_best: fp32, bf16 or APGD_I8 (not the blocked order).  Algorithmic bytes per element: 20 (16 at i = 0)
 * for the step + for a flagged sample 8 + 4 per destination written (the restore counts 8 + 8).  The LAST iteration's row moves
 * have no following step and stay with apgd_track_rows(final = 1). */
int apgd_linf_step_track_f32(const float* x, float* x_adv, const float* x_adv_old,
                             const void* grad, int grad_dtype, const float* step_size, float* out,
                             const uint8_t* flags, float* x_best, void* grad_best, float* x_best_adv,
                             int64_t B, int64_t E, float eps, float a, void* stream);

/* a8 L2 step — autopgd_train_clean.py:228-237 with L2_norm (:14-18).  Four passes
 * (three per-sample sum-of-squares reductions, wavefront __shfl + LDS, deterministic order)
 * then the final projection.  `ws` is caller-provided scratch of 3*B*apgd_l2_parts() floats
 * (per-block partial sums, combined in a fixed order).  Parity with the reference is 1e-5
 * (reduction order), not bit-exact. */
int apgd_l2_parts(void);
int apgd_l2_step_f32(const float* x, const float* x_adv, const float* x_adv_old,
                     const float* grad, const float* step_size, float* out, float* ws,
                     int64_t B, int64_t E, float eps, float a, void* stream);

/* a3/a4 per-sample loss, prediction and d(sum loss)/d logits —
 * criterion_dict['ce'] (autopgd_train_clean.py:113, 181, 275), acc/pred (:194-197, 291-294),
 * dlr_loss (:99-104) with loss_kind 1 (hard labels, n_cls >= 3; dlogits = its exact gradient).
 * logits: [B, n_cls] with row stride `ld` elements, dtype APGD_F32/BF16/F16.
 * Exactly one of y_hard (int64 [B]) / y_soft (fp32 [B, n_cls], mixup) is non-NULL.
 * loss[B] fp32;  pred[B] = (argmax(logits) == y) or (== argmax(y_soft)), first maximal index.
 * dlogits (nullable, dtype/stride of logits) = softmax*sum(y) - y  (what autograd feeds back
 * for loss_indiv.sum(), :182-185). */
int apgd_loss_pred(const void* logits, int dtype, int64_t ld,
                   const int64_t* y_hard, const float* y_soft, int loss_kind,
                   float* loss, uint8_t* pred, void* dlogits,
                   int64_t B, int64_t n_cls, void* stream);

/* a18 targeted DLR — dlr_loss_targeted (autopgd_train_clean.py:106-111), the APGD-T criterion of the
 * AutoAttack evaluation AA_eval.py drives (:226-239):
 *     loss = -(z[y] - z[y_target]) / (z_(1) - 0.5 (z_(3) + z_(4)) + 1e-12),   z_(k) = k-th largest logit.
 * Same conventions as apgd_loss_pred (pred = argmax == y; dlogits nullable = exact gradient of sum(loss));
 * y_hard, y_target int64 [B]; n_cls >= 4. */
int apgd_loss_pred_targeted(const void* logits, int dtype, int64_t ld,
                            const int64_t* y_hard, const int64_t* y_target,
                            float* loss, uint8_t* pred, void* dlogits,
                            int64_t B, int64_t n_cls, void* stream);

/* a4-a6 per-sample state machine — autopgd_train_clean.py:296, 319-324, 329-343
 * (+ check_oscillation :116-121).  One thread per sample, no host round trip:
 *   acc &= pred; loss_steps[i] = loss; new_best = loss > loss_best; loss_best = max-update;
 *   if do_check: t = sum_{c<k}[loss_steps[i-c] > loss_steps[i-c-1]] (row -1 wraps to row K-1);
 *                fl = max(t <= thr, (1-reduced_last)*(loss_best_last >= loss_best));
 *                reduced_last = fl; loss_best_last = loss_best; if fl: step_size /= 2.
 * flags[b] = NEW_BEST | MISCLS<<1 | HALVE<<2 drives apgd_track_rows_f32.
 * loss_steps is [K, B] fp32, zero-initialised by the caller (:144). thr = fp32(k*0.75) (:121). */
int apgd_state_update(const float* loss, const uint8_t* pred,
                      uint8_t* acc, float* loss_best, float* loss_best_last, float* reduced_last,
                      float* step_size, float* loss_steps, uint8_t* flags,
                      int64_t B, int32_t K, int32_t i, int32_t do_check, int32_t k, float thr,
                      void* stream);

/* a4-a6 row moves — autopgd_train_clean.py:304, 322-323, 345-346.  Per sample, by flags[b]:
 *   NEW_BEST : x_best <- x_adv, grad_best <- grad
 *   MISCLS   : x_best_adv <- x_adv
 *   HALVE    : x_adv <- x_best, grad <- grad_best   (after the NEW_BEST copy, as in the reference)
 * `final` != 0 (last iteration): only the copies that reach the return tuple are done
 * (grad is stale there, :281-283, and x_adv/grad are never read again).
 * grad / grad_best may be NULL together (no gradient tracked).  grad_elt = bytes per grad
 * element (4 fp32, 2 bf16, 1 int8 signs). */
int apgd_track_rows(const uint8_t* flags, float* x_adv, void* grad, float* x_best,
                    void* grad_best, float* x_best_adv, int32_t grad_elt,
                    int64_t B, int64_t E, int32_t final, void* stream);

/* device-side invariant (utils_eval.py:67-81 `check_imgs`): per-sample max |adv - x| and
 * min/max of adv;  out[b*3 + {0,1,2}] = {linf, min, max}.  Used by tests and --check runs. */
int apgd_check_imgs_f32(const float* adv, const float* x, float* out, int64_t B, int64_t E,
                        void* stream);

/* adv.attack = fgsm - the other value of the trainer's attack selector (main.py:836-842 -> fgsm_train.py:72-100): random start,
 * one signed gradient step, projection; two element-wise passes around the single forward / backward.
 *   apgd_fgsm_start_f32 : x_adv = x + ((2 t - 1) * eps) * noise_level, clamped to [0, 1] when clamp != 0 (:81-84); t = the uniform
 *                         draw of the reference (:81), produced by the caller with its own generator
 *   apgd_fgsm_step_f32  : out = x_adv + alpha_eps * sign(grad);  project != 0:  out = clamp01(x + clamp(out - x, -eps, eps)) (:95-98).
 *                         alpha_eps = (float)(alpha * eps) formed in double by the caller; grad fp32, bf16 or int8 signs
 * n = number of elements; every operation is rounded to fp32 on its own (no contraction), results bit-identical to the reference's. */
int apgd_fgsm_start_f32(const float* x, const float* t, float* x_adv, int64_t n, float eps, float noise_level, int32_t clamp,
                        void* stream);
int apgd_fgsm_step_f32(const float* x, const float* x_adv, const void* grad, int32_t grad_dtype, float* out, int64_t n,
                       float alpha_eps, float eps, int32_t project, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APGD_HIP_H_ */
