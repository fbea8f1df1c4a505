// block_kernels.hip — fused ConvNeXt block tail for gfx950 (MI355X):
//     out = x + gamma * ( GELU( LN(u) W1^T + b1 ) W2^T + b2 )          u = depthwise-7x7 output
// (/root/reference/models/convnext.py:40-49: norm -> pwconv1 -> GELU -> pwconv2 -> gamma -> residual).
//
// One kernel per direction; LayerNorm is the prologue, the 4C-wide hidden activation never leaves the CU.
//
// Work decomposition.  A workgroup is 4 wavefronts; each wavefront owns 32 rows of M for the whole kernel
// (its LN'd rows live in registers as MFMA operands, its 32 x C fp32 output tile lives in accumulators).
// The hidden dimension is walked in slices of 32:
//     GEMM1  Ht[32 h][32 m]  = W1[slice] (A operand, from LDS) x LN(u)^T (B operand, registers)   mfma 32x32x16 bf16
//     GELU   on the 16 accumulator registers per lane (bias pre-loaded into the accumulator), packed to bf16 —
//            the accumulator layout of Ht IS an A-operand layout of H with the k index permuted inside the slice
//     GEMM2  O[32 m][32 c]  += H (A operand, registers) x W2[slice]^T (B operand, from LDS)
// so nothing but weights goes through LDS.  Weights are pre-arranged (cnx_mlp_pack_weights) in the exact
// order the lanes read them ("fragment order": 1 KiB per MFMA operand, lane l at byte 16*l), which makes
// every ds_read_b128 linear and conflict-free without padding and lets the slices stream HBM/L2 -> LDS with
// global_load_lds (async DMA, no VGPR staging) through a 3-deep ring: one s_barrier per slice, DMA two
// slices ahead, counted vmcnt.  GEMM2 produces O (not O^T): for a fixed accumulator register the 32 lanes of
// a half-wave hold 32 consecutive channels of one row, so the epilogue (b2, gamma, residual, store) is
// coalesced without an LDS transpose.
//
// K-index conventions (any permutation of a contraction index is legal as long as both operands agree):
//   GEMM1 k = channel:  lane (row = l&31, half = l>>5) holds channels  half*C/2 + ks*8 + e  (e = 0..7) in
//                       k-step ks  -> every lane loads one contiguous half row of u (C bytes).
//   GEMM2 k = hidden:   register t*8+e of lane-half `half` is hidden unit (e&3) + 8*(2t + (e>>2)) + 4*half
//                       of the slice (the 32x32 accumulator row map).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "apgd_hip.h"
#include "convnext_hip.h"
#include "mlp_internal.h"
#include "dw_internal.h"

// Timing experiments (APGD_BLK_DBG) are compiled in only with -DMLP_ABLATE=1: tested at run time inside the hidden loop they
// become branches that split the scheduling region (see mlp_kernels.hip).
#ifndef MLP_ABLATE
#define MLP_ABLATE 0
#endif
#define DBG(p, bit) (MLP_ABLATE && ((p).dbg & (bit)))
// compile-time ablations of the pipelined loop (-DPIPE_ABL=mask: 1 no weight DMA, 2 no GELU, 4 no MFMAs, 8 no LDS fragment reads)
#ifndef PIPE_ABL
#define PIPE_ABL 0
#endif
#define PABL(bit) ((PIPE_ABL & (bit)) != 0)
// compile-time ablations of blk2_fwd_kernel (-DBLK2_ABL=mask: 1 no GELU, 2 one MFMA of GEMM1 per block, 4 one MFMA of GEMM2 per block)
#ifndef BLK2_ABL
#define BLK2_ABL 0
#endif
// issue priority inside a wavefront pair (s_setprio): 0 = none, 1 = the producer (a chain of DEPENDENT MFMAs: whenever its next one is ready
// it should go ahead of the consumer's independent ones), 2 = the consumer.  Measured in round 6 (profiles/r06_fused_mlp.md)
#ifndef BLK2_PRIO
#define BLK2_PRIO 0
#endif
// epilogue of the wavefront-pair forward: 0 = the consumer does all of it (round 5); 1 = the PRODUCER - idle from its last block on -
// loads the pair's residual tile into its dead registers under the consumer's last GEMM2 block and then does the arithmetic and the
// stores of both passes, the consumer only scatters its accumulators to LDS (round 6)
#ifndef BLK2_EPI
#define BLK2_EPI 1
#endif
#if MLP_ABLATE
// per-workgroup phase stamps (100 MHz wall clock) of the forward kernel, read back with cnx_dbg_blk_trace (tools/blk_trace.py)
#define BLK_TRACE_SLOTS 12
#define BLK_TRACE_WGS 8192
__device__ unsigned long long g_blk_trace[BLK_TRACE_WGS * BLK_TRACE_SLOTS];
#define TRACE(slot)                                                                                              \
  if (threadIdx.x == 0 && blockIdx.x < BLK_TRACE_WGS) {                                                          \
    g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + (slot)] = wall_clock64();                                         \
    if ((slot) == 1) g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 5] = __builtin_readcyclecounter();             \
    if ((slot) == 2) g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 6] = __builtin_readcyclecounter();             \
  }
#else
#define TRACE(slot)
#endif

// measurement builds only (`make EXTRA=-DBLK_FWD_W8_BUILD=1`): the eight-wavefront forward workgroups of round 6 (13 - 23 % slower than two
// independent four-wavefront workgroups per CU, profiles/r06_fused_mlp.md) are not instantiated in the product library
#ifndef BLK_FWD_W8_BUILD
#define BLK_FWD_W8_BUILD 0
#endif
#ifndef BLK_FWD_W8_DEFAULT
#define BLK_FWD_W8_DEFAULT 0
#endif

namespace {
constexpr int kFwdW8Default = BLK_FWD_W8_DEFAULT;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

// Weight DMA: one global_load_lds_dwordx4 (64 lanes x 16 bytes -> 1 KiB of LDS at `ldst`, a wave-uniform address; lane l lands at
// ldst + 16 l).  Inline asm on purpose (BLK_GLDS_ASM=0 selects the builtin for A/B): the builtin is a FLAT-encoded instruction
// with a global AND an LDS memory operand, for which the compiler's wait-count pass sets its "pending flat" state - every
// lgkmcnt / vmcnt wait it inserts while one is in flight becomes a wait for ZERO.  In the hidden loops that turned the counted
// wait in front of every third MFMA (fragment read four MFMAs ago) into lgkmcnt(0) - a wait for the fragment read issued one
// MFMA ago, ~80 cycles each, a dozen per C = 384 slice.  Completion of these loads is counted by hand (asm vmcnt waits before
// the slice barriers); compiler-inserted vmcnt waits do not know them and can only over-wait (in-order return).
#ifndef BLK_GLDS_ASM
#define BLK_GLDS_ASM 1
#endif
// The source is split into a wave-uniform base (SGPR pair: weights + slice + piece offsets, scalar arithmetic) and the lane's
// 16 l byte offset (one loop-invariant VGPR): a per-lane 64-bit source pointer cost a v_lshl_add_u64 and the scalar work to
// feed it per piece - with M0 saved and restored around every load that was ~10 instructions per KiB in loops that are bound by
// instruction issue (10 instructions per MFMA, 51 cycles per MFMA: profiles/r03_fused_mlp_issue.md).  Nothing else in these
// kernels uses M0 (no other LDS-DMA, no indexed register access), so it is set and left.
// `ldst` is the LDS BYTE ADDRESS (lds_addr(ptr) once per kernel + integer offsets: a pointer cast per piece carries a null test).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)p)); }
__device__ __forceinline__ void glds16(const unsigned char* ubase, uint32_t lane_off, uint32_t ldst) {
#if BLK_GLDS_ASM
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(ubase), "v"(lane_off), "s"(ldst) : "memory");
#else
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(ubase + lane_off), (__attribute__((address_space(3))) void*)(uintptr_t)ldst, 16, 0, 0);
#endif
}

// fp32 pair -> packed bf16 pair, round to nearest even (one v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// GELU(z) = z * Phi(z) with erfc(|z|/sqrt 2) = 2^Q(|z|), Q a degree-5 polynomial (max |error| of the
// resulting GELU 1.2e-6 over all z, fitted against scipy's erfc; tools/fit_gelu.py):
//   GELU(z) = max(z, 0) - 0.5 |z| 2^Q(|z|)             (one v_exp_f32, no division, no branch)
__device__ __forceinline__ float erfc_q(float az) {
  float q = fmaf(-0.00041175442346105595f, az, 0.006678475199902348f);
  q = fmaf(q, az, -0.050879760394516485f);
  q = fmaf(q, az, -0.46094072908550926f);
  q = fmaf(q, az, -1.150400682855232f);
  q = fmaf(q, az, -8.454223479528131e-05f);
  return __builtin_amdgcn_exp2f(q);
}
__device__ __forceinline__ float gelu_f(float z) {
  const float az = fabsf(z);
  return fmaf(az * erfc_q(az), -0.5f, fmaxf(z, 0.0f));
}
// GELU'(z) = Phi(z) + z phi(z),  Phi(z) = z > 0 ? 1 - e/2 : e/2,  phi(z) = exp(-z^2/2)/sqrt(2 pi)
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float az = fabsf(z);
  const float he = 0.5f * erfc_q(az);
  const float Phi = z > 0.0f ? 1.0f - he : he;
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * z * z);
  return fmaf(z, pdf, Phi);
}

// ---- two values per instruction.  The hidden-slice loops are VALU-bound, not MFMA-bound (measured: ~230 VALU + 32
// v_exp_f32 against 18 MFMAs per slice in the backward), so the activation math runs on v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (two fp32 lanes per instruction) and with as few quarter-rate transcendentals as possible.
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 exp2_2(f32x2 t) { return (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)}; }

// GELU of a pair (same polynomial as gelu_f), packed to bf16:  GELU(z) = 0.5 z + |z| (0.5 - 0.5 E),  E = erfc(|z|/sqrt 2).
// 13 VALU per pair; the max(z, 0) form cost 17 (two v_max plus the canonicalising v_max the compiler puts in front of an
// fmaxf on MFMA results) - every wave64 VALU instruction costs 4 cycles on this part (profiles/r02_fused_mlp_study.md).
__device__ __forceinline__ uint32_t gelu2_bf16(float z0, float z1) {
  const f32x2 z = {z0, z1};
  const f32x2 az = {fabsf(z0), fabsf(z1)};
  f32x2 q = fma2(splat2(-0.00041175442346105595f), az, splat2(0.006678475199902348f));
  q = fma2(q, az, splat2(-0.050879760394516485f));
  q = fma2(q, az, splat2(-0.46094072908550926f));
  q = fma2(q, az, splat2(-1.150400682855232f));
  q = fma2(q, az, splat2(-8.454223479528131e-05f));
  const f32x2 e = exp2_2(q);
  const f32x2 w = fma2(e, splat2(-0.5f), splat2(0.5f));
  const f32x2 g = fma2(az, w, z * splat2(0.5f));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(g, bf16x2));
}

// The same GELU in UNPACKED fp32 instructions.  On gfx950 the packed-fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, ...) execute
// on the matrix pipe's time - a SIMD does not overlap them with an MFMA, neither from the same wavefront nor from another - while
// every other VALU instruction (v_fma_f32, v_exp_f32, v_cvt_pk_bf16_f32, integer ops) hides behind a running MFMA
// (tools/probe/overlap_probe.cpp, profiles/r02_power_and_overlap.md).  19 instructions per pair instead of 13, but they run while
// the matrix pipe is busy: the pipelined hidden loop (C >= 128) places them between the MFMAs.  Inline asm because the vectoriser
// re-packs scalar fp32 chains; |z| is a VOP3 source modifier here, so there is no v_and.
__device__ __forceinline__ float fma_abs_s(float q, float z, float c) {       // q * |z| + c, c uniform
  float d;
  asm("v_fma_f32 %0, %1, |%2|, %3" : "=v"(d) : "v"(q), "v"(z), "s"(c));
  return d;
}
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// The unpacked GELU of FOUR values as a list of 38 single instructions ("micro-ops"), step-major so that consecutive instructions
// belong to different values: a dependent v_fma_f32 issues every 6 cycles, an independent one every 3.7 (overlap_probe).
//   u = 4*step + el, step 0: hz = z/2   1-5: Horner   6: exp2   7: w = 0.5 - 0.5 E   8: g = |z| w + hz;   u = 36, 37: the two bf16 pairs
__device__ __forceinline__ void gelu_uop(int u, float z0, float z1, float z2, float z3, float (&q)[4], float (&hz)[4], uint32_t& pk0,
                                         uint32_t& pk1, float c5v) {
  if (u >= 36) {
    if (u == 36) pk0 = cvt_pk_bf16(q[0], q[1]); else pk1 = cvt_pk_bf16(q[2], q[3]);
    return;
  }
  const int step = u >> 2, el = u & 3;
  const float z = el == 0 ? z0 : el == 1 ? z1 : el == 2 ? z2 : z3;
  float& t = q[el];
  switch (step) {
    case 0: asm("v_mul_f32 %0, 0.5, %1" : "=v"(hz[el]) : "v"(z)); break;
    case 1: t = fma_abs_s(c5v, z, 0.006678475199902348f); break;
    case 2: t = fma_abs_s(t, z, -0.050879760394516485f); break;
    case 3: t = fma_abs_s(t, z, -0.46094072908550926f); break;
    case 4: t = fma_abs_s(t, z, -1.150400682855232f); break;
    case 5: t = fma_abs_s(t, z, -8.454223479528131e-05f); break;
    case 6: asm("v_exp_f32 %0, %1" : "=v"(t) : "v"(t)); break;
    case 7: asm("v_fma_f32 %0, %1, -0.5, 0.5" : "=v"(t) : "v"(t)); break;
    default: asm("v_fma_f32 %0, |%1|, %2, %3" : "=v"(t) : "v"(z), "v"(t), "v"(hz[el])); break;
  }
}

// GELU'(z) with ONE exponential per value:  GELU'(-a) = 0.5 erfc(a/sqrt 2) - a phi(a) = E W(x),  E = exp(-a^2/2) = 2^(-x^2),
// x = a sqrt(log2(e)/2), W(x) = 0.5 erfcx(a/sqrt 2) - a/sqrt(2 pi) ~ degree-6 polynomial (max |error| 1.6e-5, tools/
// fit_gelu_grad.py; a = min(|z|, 6): GELU'(-6) = -3e-8), and  GELU'(z) = 0.5 + copysign(0.5 - E W, z).
// Also returns E (phi(z) = E / sqrt(2 pi)), from which GELU(z) = z (GELU'(z) - z phi(z)) costs three more instructions.
__device__ __forceinline__ f32x2 gelu_grad2(float z0, float z1, f32x2& E) {
  const f32x2 a = {fminf(fabsf(z0), 6.0f), fminf(fabsf(z1), 6.0f)};
  const f32x2 x = a * splat2(0.8493218002880191f);
  E = exp2_2(-(x * x));
  f32x2 w = fma2(splat2(1.8761737253e-03f), x, splat2(-1.8196647143e-02f));
  w = fma2(w, x, splat2(7.6242087502e-02f));
  w = fma2(w, x, splat2(-1.9087504279e-01f));
  w = fma2(w, x, splat2(3.3884271219e-01f));
  w = fma2(w, x, splat2(-9.3857446811e-01f));
  w = fma2(w, x, splat2(4.9998430368e-01f));
  const f32x2 h = splat2(0.5f) - E * w;
  return splat2(0.5f) + (f32x2){copysignf(h.x, z0), copysignf(h.y, z1)};
}
__device__ __forceinline__ f32x2 gelu_from_grad2(float z0, float z1, f32x2 gp, f32x2 E) {
  const f32x2 z = {z0, z1};
  const f32x2 Phi = fma2(z * E, splat2(-0.3989422804014327f), gp);
  return z * Phi;
}


// dHpre = dH * GELU'(Hpre) for FOUR values as 62 single UNPACKED instructions, step-major (as gelu_uop; same arithmetic and rounding
// points as gelu_grad2 above, so the results are bit-identical to the packed form):
//   u = 4*step + el, step 0: a = min(|z|, 6)  1: x = a k  2: t = -(x x)  3: E = exp2 t  4-9: W(x) Horner  10: E W  11: 0.5 - .
//   12: copysign(., z)  13: 0.5 + .  14: dH * .;   u = 60, 61: the two bf16 pairs
__device__ __forceinline__ void gelu_grad_uop(int u, const float (&z)[4], const float (&dh)[4], float (&x)[4], float (&e)[4], float (&w)[4],
                                              uint32_t& pk0, uint32_t& pk1, float c6v) {
  if (u >= 60) {
    if (u == 60) pk0 = cvt_pk_bf16(w[0], w[1]); else pk1 = cvt_pk_bf16(w[2], w[3]);
    return;
  }
  const int step = u >> 2, el = u & 3;
  switch (step) {
    case 0: asm("v_min_f32 %0, |%1|, %2" : "=v"(x[el]) : "v"(z[el]), "s"(6.0f)); break;
    case 1: asm("v_mul_f32 %0, %1, %2" : "=v"(x[el]) : "v"(x[el]), "s"(0.8493218002880191f)); break;
    case 2: asm("v_mul_f32 %0, -%1, %1" : "=v"(e[el]) : "v"(x[el])); break;
    case 3: asm("v_exp_f32 %0, %1" : "=v"(e[el]) : "v"(e[el])); break;
    case 4: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(c6v), "v"(x[el]), "s"(-1.8196647143e-02f)); break;
    case 5: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(w[el]), "v"(x[el]), "s"(7.6242087502e-02f)); break;
    case 6: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(w[el]), "v"(x[el]), "s"(-1.9087504279e-01f)); break;
    case 7: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(w[el]), "v"(x[el]), "s"(3.3884271219e-01f)); break;
    case 8: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(w[el]), "v"(x[el]), "s"(-9.3857446811e-01f)); break;
    case 9: asm("v_fma_f32 %0, %1, %2, %3" : "=v"(w[el]) : "v"(w[el]), "v"(x[el]), "s"(4.9998430368e-01f)); break;
    case 10: asm("v_mul_f32 %0, %1, %2" : "=v"(w[el]) : "v"(e[el]), "v"(w[el])); break;
    case 11: asm("v_sub_f32 %0, 0.5, %1" : "=v"(w[el]) : "v"(w[el])); break;
    case 12: asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(w[el]) : "s"(0x7fffffff), "v"(w[el]), "v"(z[el])); break;
    case 13: asm("v_add_f32 %0, 0.5, %1" : "=v"(w[el]) : "v"(w[el])); break;
    default: asm("v_mul_f32 %0, %1, %2" : "=v"(w[el]) : "v"(dh[el]), "v"(w[el])); break;
  }
}

// W: wavefronts per workgroup = 32-row tiles that share one weight stream (4: rounds 1 - 5; 8 - round 6 - halves the weight bytes a CU
// pulls through L2 -> LDS per row where the registers allow two wavefronts per SIMD: C = 128, 192)
template <int C, int W = 4>
struct Geo {
  static constexpr int KS = C / 16;                 // k-steps of a GEMM whose contraction runs over channels
  static constexpr int CB = C / 32;                 // 32-wide blocks of channels
  static constexpr int NHB = C / 8;                 // 32-wide slices of the hidden dimension (4C / 32)
  static constexpr int FWD_PIECES = KS + 2 * CB;    // 1 KiB operand fragments per slice: W1 (KS) + W2 (2 CB)
  static constexpr int FWD_SLICE = FWD_PIECES * 1024;
  static constexpr int FWD_ROUNDS = FWD_PIECES / W; // DMA instructions per wavefront per slice
#ifndef BLK_FWD96_DEPTH
#define BLK_FWD96_DEPTH 3
#endif
  static constexpr int DEPTH = (C <= 96) ? BLK_FWD96_DEPTH : 3;   // ring slots (DEPTH - 1 slices in flight ahead of the one being read)
  static constexpr int FWD_LDS = DEPTH * FWD_SLICE + 32 * C;   // + b1 (4C), b2 (C), gamma (C), ln_w (C), ln_b (C) fp32
  static_assert(FWD_LDS <= 160 * 1024, "ring + constants must fit the CU's LDS");
  static constexpr int BM = 32 * W;                 // rows per workgroup
  static constexpr int RP = (W * 16 * C * 4 <= DEPTH * FWD_SLICE) ? 16 : 8;   // rows per epilogue pass (the tile leaves through the dead ring)
  static_assert(W * RP * C * 4 <= DEPTH * FWD_SLICE, "the epilogue's passes fit the ring");
  // PIPE: the hidden loop is software-pipelined inside every wavefront - GEMM1 of hidden block t runs while the (unpacked) GELU
  // of block t-1 is evaluated - and the packed weights carry one more slice: slice t = [W1(t) | W2(t-1)], t = 0..NHB.
  static constexpr bool PIPE = blk_fwd_pipe(C);
  static constexpr int NSL = NHB + (PIPE ? 1 : 0);  // weight slices streamed through the ring
  static_assert(FWD_PIECES % W == 0, "pieces must divide over the wavefronts");
};

// ------------------------------------------------------------------ weight pre-arrangement
// Wf: [NHB][FWD_PIECES][64 lanes][8] bf16.  piece p < KS: W1 operand of k-step p; then the W2 operands, (t, cb) order.
template <typename TW>
__global__ __launch_bounds__(256) void pack_fwd_kernel(const TW* __restrict__ W1, const TW* __restrict__ W2,
                                                       uint16_t* __restrict__ Wf, int C) {
  const int KS = C / 16, PIECES = KS + 2 * (C / 32), NHB = C / 8;
  const bool pipe = blk_fwd_pipe(C);
  const long total = static_cast<long>(NHB + (pipe ? 1 : 0)) * PIECES * 64;
  const long q = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= total) return;
  const int lane = static_cast<int>(q & 63), l32 = lane & 31, half = lane >> 5;
  const int p = static_cast<int>((q >> 6) % PIECES);
  int hb = static_cast<int>((q >> 6) / PIECES);
  float v[8];
  if (pipe) {                                           // slice t = [W1(t) | W2(t-1)]; the two pieces that do not exist are zeros
    if (p >= KS) --hb;
    if (hb < 0 || hb >= NHB) {
      reinterpret_cast<uint4*>(Wf)[q] = make_uint4(0u, 0u, 0u, 0u);
      return;
    }
  }
  if (p < KS) {
    const TW* src = W1 + static_cast<long>(hb * 32 + l32) * C + half * (C / 2) + p * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = static_cast<float>(src[e]);
  } else {
    const int t = (p - KS) / (C / 32), cb = (p - KS) % (C / 32);      // t-major: consecutive MFMAs hit different accumulators
    const TW* src = W2 + static_cast<long>(cb * 32 + l32) * (4 * C) + hb * 32;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = static_cast<float>(src[(e & 3) + 8 * (2 * t + (e >> 2)) + 4 * half]);
  }
  uint4 o;
  o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
  reinterpret_cast<uint4*>(Wf)[q] = o;
}

#ifndef BLK_FWD96_OCC
#define BLK_FWD96_OCC 2
#endif
// WS: the pipelined loop also writes the Hpre workspace (cnx_block_mlp_fwd_hpre).  A template flag, not a test of p.hpre: ANY
// branch inside the hidden loop makes the compiler's wait-count pass merge its scoreboards at the join and wait lgkmcnt(0) - for
// the fragment read issued one MFMA ago - instead of the counted wait (a dozen ~80-cycle stalls per C = 384 slice in round 2's
// loop, which tested p.hpre and the "is there a slice left to prefetch" conditions at run time).
// WS == 2 (the training forward on this kernel pair, round 5): also H = GELU(Hpre) into a second workspace of the same tiles and the
// LN(u) rows - the operands of the weight-gradient contractions (cnx_gemm_tn_ex).  Workspace tile (32 rows x 32 hidden units, 2 KiB,
// tile index = (row tile, hidden block)): the lane (row m = lane % 32, half = lane / 32) stores its 16 accumulator values as bf16 at
// byte 64 m + 32 half: rows of 64 bytes = the CNX_TN_ACC layout of include/convnext_hip.h (opaque to every other caller).
template <int C, typename TX, typename TO, int WS = 0, int W = 4>
__global__ __launch_bounds__(64 * W, (W == 8 ? 2 : C <= 96 ? BLK_FWD96_OCC : C <= 192 ? 2 : 1)) void blk_mlp_fwd_kernel(const BlkFwdArgs p) {
  using G = Geo<C, W>;
  constexpr int NT = 64 * W;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ring = lds;
  float* b1s = reinterpret_cast<float*>(lds + G::DEPTH * G::FWD_SLICE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * G::BM + wave * 32;
  TRACE(0)
#if MLP_ABLATE
  if (threadIdx.x == 0 && blockIdx.x < BLK_TRACE_WGS) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 7] = (static_cast<unsigned long long>(xcc) << 32) | hw;
  }
#endif

  // ---- weight stream: slice s -> ring slot s % DEPTH, 1 KiB pieces, piece = round*W + wave
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wf);    // wave-uniform; the lane's part is lane16
  const uint32_t lane16 = lane * 16, ring0 = __builtin_amdgcn_readfirstlane(lds_addr(ring));
#define DMA_SLICE(S)                                                                                       \
  {                                                                                                        \
    const unsigned char* gs = wsrc + static_cast<long>(S) * G::FWD_SLICE;                                  \
    const uint32_t ls = ring0 + ((S) % G::DEPTH) * G::FWD_SLICE;                                           \
    _Pragma("unroll") for (int i = 0; i < G::FWD_ROUNDS; ++i) {                                            \
      const int piece = i * W + wave;                                                                      \
      glds16(gs + piece * 1024, lane16, ls + piece * 1024); \
    }                                                                                                      \
  }
#define DMA_PIECE(S, R)                                                                                    \
  {                                                                                                        \
    const unsigned char* gs = wsrc + static_cast<long>(S) * G::FWD_SLICE;                                  \
    const uint32_t ls = ring0 + ((S) % G::DEPTH) * G::FWD_SLICE;                                           \
    const int piece = (R) * W + wave;                                                                      \
    glds16(gs + piece * 1024, lane16, ls + piece * 1024); \
  }
  if (DBG(p, 16) && blockIdx.x < 1024) {                // timing experiment: de-phase the co-resident workgroups of the first round
    const int steps = ((blockIdx.x >> 8) & 3) * (p.dbg >> 8);
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue order (round 3): the per-channel constants go to LDS and this lane's half row of u into registers FIRST, the
  //      weight DMA of the first two slices is issued behind them.  The DMA loads are inline asm the compiler does not count:
  //      any wait it inserts for one of ITS loads issued after them also waits for them (in-order return) - round 2 issued the
  //      96 KB of DMA first and every wavefront sat in `vmcnt(3)` for its u row until both slices had landed, then fetched the
  //      LayerNorm weights from global memory in C/64 rounds of eight loads and a vmcnt(0) each (13.6 us of a 95 us workgroup at
  //      C = 384, tools/blk_trace.py).
  for (int i = tid; i < C; i += NT) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  for (int i = tid; i < C; i += NT) {                                   // b2, gamma (1 when absent), LayerNorm weight / bias
    b1s[4 * C + i] = p.b2[i];
    b1s[5 * C + i] = p.gamma ? p.gamma[i] : 1.0f;
    b1s[6 * C + i] = p.ln_w ? p.ln_w[i] : 1.0f;
    b1s[7 * C + i] = p.ln_w ? p.ln_b[i] : 0.0f;
  }

  // ---- this lane's half row of u  ->  (LayerNorm)  ->  GEMM1 B-operand fragments
  long row = m0 + l32;
  const bool row_ok = row < p.M;
  if (!row_ok) row = p.M - 1;
  bf16x8 af[G::KS];
  {
    uint4 raw[G::KS];
    const uint4* up = reinterpret_cast<const uint4*>(p.u + row * C + half * (C / 2));
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) raw[ks] = up[ks];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)                                  // the row is in registers before the DMA is issued
      asm volatile("" : "+v"(raw[ks].x), "+v"(raw[ks].y), "+v"(raw[ks].z), "+v"(raw[ks].w));
#pragma unroll
    for (int s0 = 0; s0 < G::DEPTH - 1; ++s0) DMA_SLICE(s0)
    __syncthreads();                                                    // constants visible (LDS only: the DMA is not waited for)
    if (p.ln_w) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bf16_lo(w[j]) + bf16_hi(w[j]);
      }
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.0f / C);
      float ss = 0.f;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16_lo(w[j]) - mean, b = bf16_hi(w[j]) - mean;
          ss = fmaf(a, a, ss);
          ss = fmaf(b, b, ss);
        }
      }
      ss += __shfl_xor(ss, 32, 64);
      const float rstd = rsqrtf(ss * (1.0f / C) + p.eps);
      if (p.mean && half == 0 && row_ok) { p.mean[row] = mean; p.rstd[row] = rstd; }
      const float4* lw = reinterpret_cast<const float4*>(b1s + 6 * C + half * (C / 2));     // LDS: a half-wave reads one address
      const float4* lb = reinterpret_cast<const float4*>(b1s + 7 * C + half * (C / 2));
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
        const float4 w0 = lw[2 * ks], w1 = lw[2 * ks + 1], c0 = lb[2 * ks], c1 = lb[2 * ks + 1];
        const float g[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float o[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = fmaf((bf16_lo(w[j]) - mean) * rstd, g[2 * j], o[2 * j]);
          const float b = fmaf((bf16_hi(w[j]) - mean) * rstd, g[2 * j + 1], o[2 * j + 1]);
          pk[j] = pack_bf16(a, b);
        }
        af[ks] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        if constexpr (WS == 2) {
          if (row_ok) reinterpret_cast<uint4*>(p.a_out + row * C + half * (C / 2))[ks] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, raw[ks]);
    }
  }

  f32x16 acc2[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;

  // ---- pull this wavefront's residual tile (32 rows x C, contiguous) towards the L2 now: one lane per 128-byte line.
  //      The epilogue's loads then pay an L2 hit instead of an HBM round trip per batch (the epilogue alone measured
  //      ~2/3 of the kernel at C = 96: latency-bound, ~4 KiB in flight per wavefront).
  constexpr int kResLines = (32 * C * static_cast<int>(sizeof(TX)) + 127) / 128;
  constexpr int kResPf = (kResLines + 63) / 64;
  uint32_t pf[kResPf];
  if (p.resid) {
    const unsigned char* rb = static_cast<const unsigned char*>(p.resid) + m0 * C * static_cast<long>(sizeof(TX));
    const long lim = (p.M - m0) * C * static_cast<long>(sizeof(TX));          // bytes of the tile that exist
#pragma unroll
    for (int i = 0; i < kResPf; ++i) {
      long off = (static_cast<long>(i) * 64 + lane) * 128;
      if (off >= lim) off = 0;
      pf[i] = (lim > 0) ? *reinterpret_cast<const uint32_t*>(rb + off) : 0u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kResPf; ++i) pf[i] = 0u;
  }

  TRACE(1)                                              // LayerNorm done, operands in registers
  if constexpr (G::PIPE) {
    // ---- software-pipelined hidden loop.  Iteration t (slice t = [W1(t) | W2(t-1)]):
    //        MFMA stream:  GEMM1(t) (KS)  ->  GEMM2(t-1) first half (CB, needs pairs 0-3 of H)  ->  GEMM2(t-1) second half (CB)
    //        VALU stream:  GELU of block t-1, 16 values per lane in UNPACKED instructions, one group after each of the first
    //                      KS + CB MFMAs - the only VALU work that runs while the matrix pipe is busy (see gelu1 above).
    //      sched_barrier(0) pins the interleaving; everything stays compiler-visible, so waits and hazards are the compiler's.
    // (PF = operand fragments in flight per wavefront; 8 instead of 4 measured the same at one wavefront per SIMD: the LDS latency
    //  is covered, what stalled the C >= 256 loops was the weight DMA - see DMA_EVERY below)
    constexpr int NF = G::KS + 2 * G::CB, PF = 4, SLOTS = G::KS + G::CB, NUOP = 4 * 38;
    // the weight DMA of slice t+2 is issued one instruction at a time between the MFMAs of iteration t: a burst of FWD_ROUNDS
    // global_load_lds at the top of the iteration stalls the in-order wavefront at issue (~1300 cycles per C = 384 slice - the
    // texture path takes a KiB per ~16 cycles and the four wavefronts of the CU share it)
    constexpr int DMA_EVERY = SLOTS / G::FWD_ROUNDS;
    static_assert(DMA_EVERY >= 1 && DMA_EVERY * (G::FWD_ROUNDS - 1) < SLOTS, "one DMA instruction per DMA_EVERY MFMA slots");
    float c5v = -0.00041175442346105595f;               // leading GELU coefficient in a VGPR (one constant-bus operand per VOP3)
    asm volatile("" : "+v"(c5v));
#if MLP_ABLATE
    unsigned long long tr_wait = 0, tr_bar = 0, tr_work = 0, tr_t = __builtin_readcyclecounter();
#define TR_ACC(ACC) { const unsigned long long n_ = __builtin_readcyclecounter(); ACC += n_ - tr_t; tr_t = n_; }
#else
#define TR_ACC(ACC)
#endif
    // ST ("steady"): a compile-time promise that slices T + 1 and T + DEPTH - 1 exist, so the iteration is straight-line code;
    // the last two iterations are instantiated with a constant T instead and their conditions fold away as well.
#define SLICE_SYNC(T, ST)                                                                                      \
    TR_ACC(tr_work)                                                                                            \
    if (ST || (T) + 1 < G::NSL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::FWD_ROUNDS) : "memory");           \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    TR_ACC(tr_wait)                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                              \
    TR_ACC(tr_bar)
#define LOAD_BIAS(Z, T)                                                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + (T) * 32 + 8 * g + 4 * half);                   \
      Z[4 * g + 0] = b4.x; Z[4 * g + 1] = b4.y; Z[4 * g + 2] = b4.z; Z[4 * g + 3] = b4.w;                      \
    }
    // (Measured and dropped in round 3: GEMM1's k-steps alternating between two accumulators, summed behind GEMM2 - the theory
    //  being that a dependent MFMA behind interleaved VALU work waits for its predecessor's write-back.  201 -> 212 us at
    //  C = 384, 232 -> 243 at C = 256: the stalls were the wait-count pass's lgkmcnt(0), see glds16.)
    f32x16 za, zb;
    {                                                   // t = 0: GEMM1 of block 0 only
      SLICE_SYNC(0, false)
      if (G::DEPTH - 1 < G::NSL) DMA_SLICE(G::DEPTH - 1)
      const unsigned char* sl = ring + lane * 16;
      bf16x8 fr[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);
      LOAD_BIAS(za, 0)
#pragma unroll
      for (int i = 0; i < G::KS; ++i) {
        za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i], za, 0, 0, 0);
        if (i + PF < G::KS) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      }
    }
#define PIPE_ITER(T, ZIN, ZOUT, ST)                                                                            \
    {                                                                                                          \
      SLICE_SYNC(T, ST)                                                                                        \
      if constexpr (WS) {   /* Hpre of block T-1 for the input-gradient kernel: 16 bf16 per lane, accumulator order */ \
        uint4* dst = reinterpret_cast<uint4*>(p.hpre) +                                                        \
                     ((static_cast<long>(blockIdx.x) * W + wave) * G::NHB + ((T) - 1)) * 128 + l32 * 4 + half * 2; \
        dst[0] = make_uint4(cvt_pk_bf16(ZIN[0], ZIN[1]), cvt_pk_bf16(ZIN[2], ZIN[3]), cvt_pk_bf16(ZIN[4], ZIN[5]),   \
                            cvt_pk_bf16(ZIN[6], ZIN[7]));                                                      \
        dst[1] = make_uint4(cvt_pk_bf16(ZIN[8], ZIN[9]), cvt_pk_bf16(ZIN[10], ZIN[11]), cvt_pk_bf16(ZIN[12], ZIN[13]), \
                            cvt_pk_bf16(ZIN[14], ZIN[15]));                                                    \
      }                                                                                                        \
      const unsigned char* sl = ring + ((T) % G::DEPTH) * G::FWD_SLICE + lane * 16;                            \
      bf16x8 fr[PF];                                                                                           \
      _Pragma("unroll") for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);  \
      LOAD_BIAS(ZOUT, T)                                                                                       \
      float gq[4], ghz[4];                                                                                     \
      uint32_t pk[8];                                                                                          \
      if (PABL(2)) { _Pragma("unroll") for (int e = 0; e < 8; ++e) pk[e] = __builtin_bit_cast(uint32_t, ZIN[2 * e]); } \
      bf16x8 hf0, hf1;                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int i = 0; i < SLOTS; ++i) {                                                      \
        if (i == G::KS) hf0 = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));              \
        if (PABL(4)) { asm volatile("" ::"v"(fr[i % PF])); }                                                  \
        else if (i < G::KS) ZOUT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i < G::KS ? i : 0], ZOUT, 0, 0, 0); \
        else acc2[(i - G::KS) % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf0, fr[i % PF], acc2[(i - G::KS) % G::CB], 0, 0, 0); \
        if (i + PF < NF && !PABL(8)) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);   \
        if (i % DMA_EVERY == 0 && i / DMA_EVERY < G::FWD_ROUNDS && !PABL(1)) {                               \
          if (ST || (T) + G::DEPTH - 1 < G::NSL) DMA_PIECE((T) + G::DEPTH - 1, i / DMA_EVERY)                  \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!PABL(2))                                                                                        \
        _Pragma("unroll") for (int uo = NUOP * i / SLOTS; uo < NUOP * (i + 1) / SLOTS; ++uo) {                 \
          const int qd = uo / 38;                                                                              \
          gelu_uop(uo % 38, ZIN[4 * qd], ZIN[4 * qd + 1], ZIN[4 * qd + 2], ZIN[4 * qd + 3], gq, ghz, pk[2 * qd], pk[2 * qd + 1], c5v); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      }                                                                                                        \
      hf1 = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));                                \
      if constexpr (WS == 2) {   /* H of block T-1 (the GEMM2 operand pairs just formed), same tile as its Hpre */ \
        uint4* hdst = reinterpret_cast<uint4*>(p.hact) +                                                       \
                      ((static_cast<long>(blockIdx.x) * W + wave) * G::NHB + ((T) - 1)) * 128 + l32 * 4 + half * 2; \
        hdst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);                                                      \
        hdst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);                                                      \
      }                                                                                                        \
      _Pragma("unroll") for (int j = G::CB; j < 2 * G::CB; ++j) {                                              \
        const int i = G::KS + j;                                                                               \
        if (PABL(4)) { asm volatile("" ::"v"(fr[i % PF]), "v"(hf1)); }                                        \
        else acc2[j - G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf1, fr[i % PF], acc2[j - G::CB], 0, 0, 0);  \
        if (i + PF < NF && !PABL(8)) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);   \
      }                                                                                                        \
    }
    static_assert(G::NHB % 2 == 0 && NUOP * G::KS / SLOTS >= 76, "pipelined loop: pairs 0-3 are ready when GEMM2 starts");
    static_assert(G::NHB >= 4 && G::NSL == G::NHB + 1 && G::DEPTH == 3, "steady iterations 1 .. NHB-2, then the constant-T tail");
    for (int t = 1; t + 1 <= G::NHB - 2; t += 2) {
      PIPE_ITER(t, za, zb, true)
      PIPE_ITER(t + 1, zb, za, true)
    }
    PIPE_ITER(G::NHB - 1, za, zb, false)
    PIPE_ITER(G::NHB, zb, za, false)
#if MLP_ABLATE
    TR_ACC(tr_work)
    if (threadIdx.x == 0 && blockIdx.x < BLK_TRACE_WGS) {
      g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 8] = tr_wait;
      g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 9] = tr_bar;
      g_blk_trace[blockIdx.x * BLK_TRACE_SLOTS + 10] = tr_work;
    }
#endif
#undef TR_ACC
#undef PIPE_ITER
#undef LOAD_BIAS
#undef SLICE_SYNC
#undef DMA_PIECE
  }
  // ---- hidden-slice loop
  const int n_slices = (G::PIPE || DBG(p, 8)) ? 0 : G::NHB;          // dbg 8: prologue + epilogue only
  for (int s = 0; s < n_slices; ++s) {
    // slice s has landed once this wavefront's own pieces are in (counted wait: slice s+1 may stay in flight) and
    // every wavefront has passed the barrier; the slot of slice s+2 held slice s-1, which nobody reads any more
    if (!DBG(p, 4)) {
      if (G::DEPTH > 3 && s + 2 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::FWD_ROUNDS) : "memory");
      else if (s + 1 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::FWD_ROUNDS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (s + G::DEPTH - 1 < G::NHB && !DBG(p, 1)) DMA_SLICE(s + G::DEPTH - 1)
    const unsigned char* sl = ring + (s % G::DEPTH) * G::FWD_SLICE + lane * 16;

    // One stream of NF = KS + 2 CB operand fragments per slice (1 KiB each, fragment i feeds MFMA i).  The reads run
    // PF fragments ahead of the MFMAs so that the LDS latency (~100+ cycles) hides behind the 32-cycle MFMAs instead of
    // serialising with them (a compiler-scheduled read-wait-MFMA chain measured ~110 cycles per MFMA).
    constexpr int NF = G::KS + 2 * G::CB, PF = 4;
    bf16x8 fr[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);
    // GEMM1, accumulator pre-loaded with b1 (register r <-> hidden (r&3) + 8*(r>>2) + 4*half)
    f32x16 acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + s * 32 + 8 * g + 4 * half);
      acc1[4 * g + 0] = b4.x; acc1[4 * g + 1] = b4.y; acc1[4 * g + 2] = b4.z; acc1[4 * g + 3] = b4.w;
    }
    // two interleaved accumulation chains: a single chain would issue each MFMA only after the previous one has
    // written back (dependent-accumulator latency > issue interval)
    f32x16 acc1b;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1b[r] = 0.f;
#pragma unroll
    for (int i = 0; i < G::KS; ++i) {
      if (i & 1) acc1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i], acc1b, 0, 0, 0);
      else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i], acc1, 0, 0, 0);
      if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] += acc1b[r];
    // GELU -> bf16 A-operand fragments of GEMM2
    bf16x8 hf[2];
    {
      uint32_t pk[8];
      if DBG(p, 2) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16(acc1[r], acc1[r + 1]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = gelu2_bf16(acc1[r], acc1[r + 1]);
      }
      hf[0] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      hf[1] = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
    }
    // GEMM2
#pragma unroll
    for (int j = 0; j < 2 * G::CB; ++j) {                      // fragment order in the slice is (t, cb): j = t*CB + cb
      const int i = G::KS + j;
      acc2[j % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[j / G::CB], fr[i % PF], acc2[j % G::CB], 0, 0, 0);
      if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  }
#undef DMA_SLICE

  // ---- epilogue: acc2[cb][r] = O[m0 + (r&3) + 8*(r>>2) + 4*half][cb*32 + l32].  Straight from the accumulators every
  //      load / store would move 4 bytes per lane (96 memory instructions per wavefront at C = 96); instead the tile goes
  //      through the (now dead) weight ring in two passes of 16 rows - registers 0..7 of every accumulator are rows 0..15 -
  //      and leaves as 16 rows x C contiguous elements: 16 bytes per lane for the residual read and the result.
#pragma unroll
  for (int i = 0; i < kResPf; ++i) asm volatile("" ::"v"(pf[i]));       // the prefetch loads are complete (and were not dropped)
  TRACE(2)                                                              // this wavefront's hidden loop is done
  __syncthreads();                                                      // every wavefront is past its last fragment read
  TRACE(3)
  constexpr int RP = G::RP;                                             // rows per pass: 16 (registers 8 pass .. 8 pass + 7), or 8 (4 pass .. 4 pass + 3)
  float* scr = reinterpret_cast<float*>(ring) + wave * (RP * C);        // RP rows x C fp32 per wavefront
  const float4* b2v = reinterpret_cast<const float4*>(b1s + 4 * C);
  const float4* gav = reinterpret_cast<const float4*>(b1s + 5 * C);
  const TX* resid = static_cast<const TX*>(p.resid);
  TO* out = static_cast<TO*>(p.out);
  constexpr int C4 = C / 4;                                             // float4 chunks per row
  constexpr int NCH = RP * C4 / 64;                                     // chunks per lane and pass
  constexpr int GRP = (NCH % 6 == 0) ? 6 : 4;                           // chunks in flight per lane
  static_assert(NCH % GRP == 0, "chunk groups");
#pragma unroll
  for (int pass = 0; pass < 32 / RP; ++pass) {
    const long e0 = (m0 + RP * pass) * C;                               // first element of this pass
    const long e_end = p.M * C;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
      for (int r = 0; r < RP / 2; ++r)
        scr[((r & 3) + 8 * (r >> 2) + 4 * half) * C + cb * 32 + l32] = acc2[cb][(RP / 2) * pass + r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g0 = 0; g0 < NCH; g0 += GRP) {
      float4 xv[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int idx = (g0 + j) * 64 + lane;
        const long e = e0 + idx * 4;
        xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (resid && e < e_end) {
          if constexpr (sizeof(TX) == 4) {
            xv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resid) + e);
          } else {
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(resid) + e);
            xv[j] = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int idx = (g0 + j) * 64 + lane;
        const long e = e0 + idx * 4;
        const int c4 = idx % C4;
        const float4 o = reinterpret_cast<const float4*>(scr)[idx];
        const float4 bb = b2v[c4], gg = gav[c4];
        const float y0 = o.x + bb.x, y1 = o.y + bb.y, y2v = o.z + bb.z, y3 = o.w + bb.w;
        if (e < e_end) {
          if (p.y2) *reinterpret_cast<uint2*>(p.y2 + e) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2v, y3));
          const float o0 = fmaf(y0, gg.x, xv[j].x), o1 = fmaf(y1, gg.y, xv[j].y);
          const float o2 = fmaf(y2v, gg.z, xv[j].z), o3 = fmaf(y3, gg.w, xv[j].w);
          if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + e) = make_float4(o0, o1, o2, o3);
          else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + e) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
        }
      }
    }
  }
  TRACE(4)                                              // stores issued (not necessarily landed)
}

// =====================================================================================================================
// The same forward with the two GEMMs of a row tile on TWO wavefronts of one SIMD (round 5).
//
// blk_mlp_fwd_kernel puts a row tile's whole chain - GEMM1, GELU, GEMM2 - on one wavefront: at C >= 256 its 32 x C fp32 output tile
// plus the LN'd operand rows fill the register file, the SIMD hosts ONE wavefront, and a wavefront issues one instruction per four
// cycles: 9 - 10 instructions per 32-cycle MFMA leave the matrix pipe 60 % busy inside the loop (profiles/r03_fused_mlp_issue.md).
// Here a workgroup is eight wavefronts = four PAIRS sharing a SIMD:
//   producer (wavefronts 0-3)  LN prologue; per hidden block s: Hpre^T = W1[s] x LN(u)^T (KS MFMAs, one accumulator), + b1, GELU in
//                              unpacked VALU instructions, H(s) as bf16 operand pairs -> 2 KiB of LDS (the accumulator layout IS the
//                              A-operand layout: the consumer lane reads back what the producer lane wrote, 32 bytes each);
//   consumer (wavefronts 4-7)  O += H(s-1) x W2[s-1]^T (2 CB MFMAs into the 32 x C fp32 tile), epilogue (b2, gamma, residual, store).
// Neither role needs more than 256 registers, so both live on the SIMD and their instruction streams issue side by side: the
// producer's ~150 VALU instructions per block run under the consumer's MFMAs instead of between a single wavefront's.  One barrier
// per hidden block hands H(s) over and recycles the weight rings.  Packed weights: the pipelined order of cnx_mlp_pack_weights
// (slice t = [W1(t) | W2(t-1)]); LDS: W1 ring 3 x KS KiB (two blocks ahead), W2 ring 2 x 2 CB KiB (one block ahead), 16 KiB of H
// hand-over buffers (two per pair), b1.  LayerNorm weights use the H buffers before the loop, b2 / gamma after it; the output tile
// leaves through the dead rings as in blk_mlp_fwd_kernel.  WS as there (1: Hpre workspace, 2: + H workspace and LN(u) rows).
template <int C>
struct Geo2 {
  static constexpr int KS = C / 16, CB = C / 32, NHB = C / 8;
  static constexpr int PIECES = KS + 2 * CB, SLICE = PIECES * 1024;
  static constexpr int R1 = KS / 8, R2 = 2 * CB / 8;            // DMA instructions per wavefront and block: W1 / W2 pieces
  static_assert(KS % 8 == 0 && (2 * CB) % 8 == 0, "pieces deal evenly over eight wavefronts (C a multiple of 128)");
  static constexpr int W1_RING = 3 * KS * 1024, W2_RING = 2 * 2 * CB * 1024, HBUF = 4 * 2 * 4096;   // per pair, two buffers of [H | Hpre] tiles
  static constexpr int LDS = W1_RING + W2_RING + HBUF + 16 * C;
  static_assert(LDS <= 160 * 1024 && 2 * C * 4 <= HBUF && 4 * 16 * C * 4 <= W1_RING + W2_RING, "LDS plan");
};

template <int C, typename TX, typename TO, int WS>
__global__ __launch_bounds__(512, 2) void blk2_fwd_kernel(const BlkFwdArgs p) {
  using G = Geo2<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* hbuf = lds + G::W1_RING + G::W2_RING;
  float* b1s = reinterpret_cast<float*>(hbuf + G::HBUF);
  float* cst = reinterpret_cast<float*>(hbuf);                        // [2C] floats: ln_w | ln_b before the loop, b2 | gamma after it
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3;
  const int l32 = lane & 31, half = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * 128 + pair * 32;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wf);
  const uint32_t lane16 = lane * 16, ring1 = __builtin_amdgcn_readfirstlane(lds_addr(lds)), ring2 = ring1 + G::W1_RING;
  // W1(t): slice t, pieces [0, KS) -> ring slot t % 3;  W2(t): slice t + 1, pieces [KS, KS + 2 CB) -> ring slot t % 2
  // (every wavefront moves an eighth of the pieces - wavefront w pieces w, w + 8, ... - one instruction per DMA_EVERY MFMAs: an
  //  LDS-DMA instruction costs its issuer 60 - 185 cycles (MI355X_MICROARCH.md), twelve of them on one role were that role's block)
#define DMA_W1_PIECE(T, I)                                                                                 \
  {                                                                                                        \
    const int q_ = (I) * 8 + wave;                                                                         \
    glds16(wsrc + static_cast<long>(T) * G::SLICE + q_ * 1024, lane16, ring1 + ((T) % 3) * (G::KS * 1024) + q_ * 1024); \
  }
#define DMA_W2_PIECE(T, I)                                                                                 \
  {                                                                                                        \
    const int q_ = (I) * 8 + wave;                                                                         \
    glds16(wsrc + static_cast<long>((T) + 1) * G::SLICE + (G::KS + q_) * 1024, lane16, ring2 + ((T) % 2) * (2 * G::CB * 1024) + q_ * 1024); \
  }
  // block B, DMA instruction K of this wavefront: first its W2(B - 1) pieces (read in block B + 1; the slot W2(B - 3) left), then its
  // W1(B + 2) pieces (read in block B + 2; the slot W1(B - 1) left)
  constexpr int R1W = G::KS / 8, R2W = 2 * G::CB / 8, NDMA = R1W + R2W;
#define BLK_DMA(B, K, ST)                                                                                  \
  if ((K) < R2W) { if (ST || ((B) >= 1 && (B) - 1 < G::NHB)) DMA_W2_PIECE((B) - 1, (K)) }                  \
  else { if (ST || (B) + 2 < G::NHB) DMA_W1_PIECE((B) + 2, (K) - R2W) }
  // ... and the end of a block: everything but this block's W1 pieces (and NST stores issued behind them) is in; barrier
#define BLK_SYNC(B, ST, NST)                                                                               \
  if (ST || (B) + 2 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R1W + (NST)) : "memory");           \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < C; i += 512) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  for (int i = tid; i < C; i += 512) { cst[i] = p.ln_w ? p.ln_w[i] : 1.0f; cst[C + i] = p.ln_w ? p.ln_b[i] : 0.0f; }
  const long tile = static_cast<long>(blockIdx.x) * 4 + pair;
  unsigned char* hb_lane = hbuf + pair * 8192 + lane * 32;           // buffer i at + 4096 i: H tile, then (+ 2048) the Hpre tile

  if (wave < 4) {
    // ================================================================ producer: LN, GEMM1, GELU, H -> LDS
    if constexpr (BLK2_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    // (its own loop: the register allocation is per kernel, and a loop shared with the consumer would keep the producer's operand
    //  rows AND the consumer's accumulators live through it - 96 + 192 registers at C = 384)
    long row = m0 + l32;
    const bool row_ok = row < p.M;
    if (!row_ok) row = p.M - 1;
    bf16x8 af[G::KS];
    {
      uint4 raw[G::KS];
      const uint4* up = reinterpret_cast<const uint4*>(p.u + row * C + half * (C / 2));
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) raw[ks] = up[ks];
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(raw[ks].x), "+v"(raw[ks].y), "+v"(raw[ks].z), "+v"(raw[ks].w));
#pragma unroll
      for (int i = 0; i < R1W; ++i) DMA_W1_PIECE(0, i)
#pragma unroll
      for (int i = 0; i < R1W; ++i) DMA_W1_PIECE(1, i)
      __syncthreads();                                                // constants visible
      if (p.ln_w) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) s += bf16_lo(w[j]) + bf16_hi(w[j]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        // (the three passes each unpack the row again: kept across the passes the 8 KS fp32 values would sit next to the KS packed
        //  quads and the operand fragments - more than the 256 registers of a wavefront that shares its SIMD)
#define RAW_OPAQUE _Pragma("unroll") for (int ks_ = 0; ks_ < G::KS; ++ks_) asm volatile("" : "+v"(raw[ks_].x), "+v"(raw[ks_].y), "+v"(raw[ks_].z), "+v"(raw[ks_].w));
        RAW_OPAQUE
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = bf16_lo(w[j]) - mean, b = bf16_hi(w[j]) - mean;
            ss = fmaf(a, a, ss);
            ss = fmaf(b, b, ss);
          }
        }
        ss += __shfl_xor(ss, 32, 64);
        const float rstd = rsqrtf(ss * (1.0f / C) + p.eps);
        if (p.mean && half == 0 && row_ok) { p.mean[row] = mean; p.rstd[row] = rstd; }
        RAW_OPAQUE
#undef RAW_OPAQUE
        const float4* lw = reinterpret_cast<const float4*>(cst + half * (C / 2));
        const float4* lb = reinterpret_cast<const float4*>(cst + C + half * (C / 2));
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
          const float4 w0 = lw[2 * ks], w1 = lw[2 * ks + 1], c0 = lb[2 * ks], c1 = lb[2 * ks + 1];
          const float g[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          const float o[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          uint32_t pk[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = fmaf((bf16_lo(w[j]) - mean) * rstd, g[2 * j], o[2 * j]);
            const float b = fmaf((bf16_hi(w[j]) - mean) * rstd, g[2 * j + 1], o[2 * j + 1]);
            pk[j] = pack_bf16(a, b);
          }
          af[ks] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
          if constexpr (WS == 2) {
            if (row_ok) reinterpret_cast<uint4*>(p.a_out + row * C + half * (C / 2))[ks] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, raw[ks]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // W1(0), W1(1): this wavefront's pieces (and its LN(u) row stores)
    __syncthreads();                                                  // ... everybody's; LN constants consumed
    float c5v = -0.00041175442346105595f;
    asm volatile("" : "+v"(c5v));
    // Block b:  MFMA stream  Hpre(b) = W1[b] x LN(u)^T + b1 (KS MFMAs, b < NHB)
    //           VALU stream  GELU of Hpre(b - 1) in unpacked instructions between those MFMAs (b >= 1); H(b - 1) -> hand-over buffer
    // ST: compile-time promise 1 <= b < NHB (straight-line code, no branch inside the block)
    constexpr int PF = 4, NUOP = 4 * 38;
#define P_BLOCK(B, ZCUR, ZPREV, ST)                                                                        \
    {                                                                                                      \
      uint32_t pk[8];                                                                                      \
      float gq[4], ghz[4];                                                                                 \
      if (ST || (B) >= 1) {                                                                                \
        if constexpr (WS >= 1) {   /* Hpre(B - 1) for the workspace: handed to the consumer next to H (it makes the global stores) */ \
          uint4* dst = reinterpret_cast<uint4*>(hb_lane + (((B) - 1) & 1) * 4096 + 2048);                  \
          dst[0] = make_uint4(cvt_pk_bf16(ZPREV[0], ZPREV[1]), cvt_pk_bf16(ZPREV[2], ZPREV[3]), cvt_pk_bf16(ZPREV[4], ZPREV[5]), cvt_pk_bf16(ZPREV[6], ZPREV[7])); \
          dst[1] = make_uint4(cvt_pk_bf16(ZPREV[8], ZPREV[9]), cvt_pk_bf16(ZPREV[10], ZPREV[11]), cvt_pk_bf16(ZPREV[12], ZPREV[13]), cvt_pk_bf16(ZPREV[14], ZPREV[15])); \
        }                                                                                                  \
      }                                                                                                    \
      if (ST || (B) < G::NHB) {                                                                            \
        const unsigned char* sl = lds + ((B) % 3) * (G::KS * 1024) + lane * 16;                            \
        bf16x8 fr[PF];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024); \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
          const float4 b4 = *reinterpret_cast<const float4*>(b1s + (B) * 32 + 8 * g + 4 * half);           \
          ZCUR[4 * g + 0] = b4.x; ZCUR[4 * g + 1] = b4.y; ZCUR[4 * g + 2] = b4.z; ZCUR[4 * g + 3] = b4.w;  \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int i = 0; i < G::KS; ++i) {                                                \
          if constexpr ((BLK2_ABL & 2) == 0)                                                               \
            ZCUR = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i], ZCUR, 0, 0, 0);              \
          else if (i == 0) ZCUR = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i], ZCUR, 0, 0, 0); \
          if (i + PF < G::KS) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);         \
          if (i % P_DMA_EVERY == 0 && i / P_DMA_EVERY < NDMA) { BLK_DMA(B, i / P_DMA_EVERY, ST) }          \
          __builtin_amdgcn_sched_barrier(0);                                                               \
          if ((ST || (B) >= 1) && (BLK2_ABL & 1) == 0) {                                                   \
            _Pragma("unroll") for (int uo = NUOP * i / G::KS; uo < NUOP * (i + 1) / G::KS; ++uo) {         \
              const int qd = uo / 38;                                                                      \
              gelu_uop(uo % 38, ZPREV[4 * qd], ZPREV[4 * qd + 1], ZPREV[4 * qd + 2], ZPREV[4 * qd + 3], gq, ghz, pk[2 * qd], pk[2 * qd + 1], c5v); \
            }                                                                                              \
          }                                                                                                \
          __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                  \
        /* the chain's result is read by inline-asm VALU instructions in the next block (cvt_pk_bf16, gelu_uop), whose operands the \
           compiler's hazard recognizer does not take for VALU reads of a matrix result: the wait states by hand */ \
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(ZCUR));                                                  \
      } else {                                                                                             \
        _Pragma("unroll") for (int k_ = 0; k_ < NDMA; ++k_) { BLK_DMA(B, k_, false) }                      \
      }                                                                                                    \
      if (!(ST || (B) < G::NHB) && (B) >= 1 && (BLK2_ABL & 1) == 0) {                                      \
        _Pragma("unroll") for (int qd = 0; qd < 4; ++qd) {                                                 \
          _Pragma("unroll") for (int uo = 0; uo < 38; ++uo)                                                \
            gelu_uop(uo, ZPREV[4 * qd], ZPREV[4 * qd + 1], ZPREV[4 * qd + 2], ZPREV[4 * qd + 3], gq, ghz, pk[2 * qd], pk[2 * qd + 1], c5v); \
        }                                                                                                  \
      }                                                                                                    \
      if (ST || (B) >= 1) {                                                                                \
        if constexpr ((BLK2_ABL & 1) != 0) { _Pragma("unroll") for (int e = 0; e < 8; ++e) pk[e] = cvt_pk_bf16(ZPREV[2 * e], ZPREV[2 * e + 1]); } \
        uint4* hw = reinterpret_cast<uint4*>(hb_lane + (((B) - 1) & 1) * 4096);                            \
        hw[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);                                                    \
        hw[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);                                                    \
      }                                                                                                    \
      BLK_SYNC(B, ST, 0)                                                                                   \
    }
    static_assert(G::NHB % 2 == 0 && G::NHB >= 6, "two-block unroll, steady blocks 1 .. NHB - 3");
    constexpr int P_DMA_EVERY = G::KS / NDMA;
    static_assert(P_DMA_EVERY >= 1 && P_DMA_EVERY * NDMA <= G::KS, "one DMA instruction per P_DMA_EVERY MFMAs");
    f32x16 za, zb;
    P_BLOCK(0, za, zb, false)
    for (int b = 1; b + 1 < G::NHB - 2; b += 2) {                     // blocks 1 .. NHB - 4 (pairs), all conditions true
      P_BLOCK(b, zb, za, true)
      P_BLOCK(b + 1, za, zb, true)
    }
    P_BLOCK(G::NHB - 3, zb, za, true)
    P_BLOCK(G::NHB - 2, za, zb, false)
    P_BLOCK(G::NHB - 1, zb, za, false)
    P_BLOCK(G::NHB, za, zb, false)
#if BLK2_EPI
    // the pair's residual tile (32 rows x C, both passes of the epilogue) into this wavefront's registers - its operand rows and
    // accumulators are dead - while the consumer runs its last GEMM2 block: the epilogue then starts with the data on chip instead of
    // with 2 x NCH / GRP dependent round trips to HBM
    constexpr int EC4 = C / 4, ENCH = 16 * EC4 / 64;
    typedef typename std::conditional<sizeof(TX) == 4, float4, uint2>::type XT;
    XT xr[2][ENCH];
    {
      const TX* resid = static_cast<const TX*>(p.resid);
      const long e_end = p.M * C;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int k = 0; k < ENCH; ++k) {
          const long e = (m0 + 16 * pass) * C + (k * 64 + lane) * 4;
          if constexpr (sizeof(TX) == 4) {
            xr[pass][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (resid && e < e_end) xr[pass][k] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resid) + e);
          } else {
            xr[pass][k] = make_uint2(0u, 0u);
            if (resid && e < e_end) xr[pass][k] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(resid) + e);
          }
        }
    }
#endif
    __builtin_amdgcn_s_barrier();                                     // block NHB + 1: the consumers' last GEMM2
#undef P_BLOCK
    for (int i = tid; i < C; i += 256) { cst[i] = p.b2[i]; cst[C + i] = p.gamma ? p.gamma[i] : 1.0f; }   // b2 | gamma for the epilogue
    __syncthreads();
#if BLK2_EPI
    {
      const float* scr = reinterpret_cast<const float*>(lds) + pair * (16 * C);
      const float4* b2v = reinterpret_cast<const float4*>(cst);
      const float4* gav = reinterpret_cast<const float4*>(cst + C);
      TO* out = static_cast<TO*>(p.out);
      const long e_end = p.M * C;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                              // the consumer has scattered this pass's 16 rows
#pragma unroll
        for (int k = 0; k < ENCH; ++k) {
          const int idx = k * 64 + lane;
          const long e = (m0 + 16 * pass) * C + idx * 4;
          const int c4 = idx % EC4;
          const float4 o = reinterpret_cast<const float4*>(scr)[idx];
          const float4 bb = b2v[c4], gg = gav[c4];
          float4 xv;
          if constexpr (sizeof(TX) == 4) xv = xr[pass][k];
          else xv = make_float4(bf16_lo(xr[pass][k].x), bf16_hi(xr[pass][k].x), bf16_lo(xr[pass][k].y), bf16_hi(xr[pass][k].y));
          const float y0 = o.x + bb.x, y1 = o.y + bb.y, y2v = o.z + bb.z, y3 = o.w + bb.w;
          if (e < e_end) {
            if (p.y2) *reinterpret_cast<uint2*>(p.y2 + e) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2v, y3));
            const float o0 = fmaf(y0, gg.x, xv.x), o1 = fmaf(y1, gg.y, xv.y);
            const float o2 = fmaf(y2v, gg.z, xv.z), o3 = fmaf(y3, gg.w, xv.w);
            if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + e) = make_float4(o0, o1, o2, o3);
            else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + e) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
          }
        }
        if (pass == 0) __syncthreads();                               // the scratch rows are free for the second pass
      }
    }
#endif
    return;
  }

  // ================================================================== consumer: weight DMA, GEMM2, epilogue
  if constexpr (BLK2_PRIO == 2) __builtin_amdgcn_s_setprio(2);
  f32x16 acc2[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
#pragma unroll
  for (int i = 0; i < R1W; ++i) DMA_W1_PIECE(0, i)
#pragma unroll
  for (int i = 0; i < R1W; ++i) DMA_W1_PIECE(1, i)
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Block b:  O += H(b - 2) x W2[b - 2]^T (2 CB MFMAs, b >= 2); between them this wavefront's share of the weight DMA
  constexpr int PFC = 4, DMA_EVERY = 2 * G::CB / NDMA;
  static_assert(DMA_EVERY >= 1 && DMA_EVERY * NDMA <= 2 * G::CB, "one DMA instruction per DMA_EVERY MFMAs");
#define C_BLOCK(B, ST)                                                                                     \
  {                                                                                                        \
    if (ST || ((B) >= 2 && (B) - 2 < G::NHB)) {                                                            \
      const uint4* hr = reinterpret_cast<const uint4*>(hb_lane + (((B) - 2) & 1) * 4096);                  \
      const uint4 hq0 = hr[0], hq1 = hr[1];                                                                \
      const bf16x8 hf0 = __builtin_bit_cast(bf16x8, hq0), hf1 = __builtin_bit_cast(bf16x8, hq1);           \
      if constexpr (WS >= 1) {   /* the workspace stores of block B - 2, in front of this block's DMA: the producer's stream stays \
                                    free of global stores (an in-order wavefront pays ~100+ cycles of issue per KiB stored) */ \
        const long tq = (tile * G::NHB + ((B) - 2)) * 128 + l32 * 4 + half * 2;                            \
        uint4* dst = reinterpret_cast<uint4*>(p.hpre) + tq;                                                \
        dst[0] = hr[128]; dst[1] = hr[129];                                                                \
        if constexpr (WS == 2) {                                                                           \
          uint4* hdst = reinterpret_cast<uint4*>(p.hact) + tq;                                             \
          hdst[0] = hq0; hdst[1] = hq1;                                                                    \
        }                                                                                                  \
      }                                                                                                    \
      const unsigned char* sl = lds + G::W1_RING + (((B) - 2) % 2) * (2 * G::CB * 1024) + lane * 16;       \
      bf16x8 fr[PFC];                                                                                      \
      _Pragma("unroll") for (int j = 0; j < PFC; ++j) fr[j] = *reinterpret_cast<const bf16x8*>(sl + j * 1024); \
      _Pragma("unroll") for (int j = 0; j < 2 * G::CB; ++j) {                                              \
        if ((BLK2_ABL & 4) == 0 || j == 0)                                                                 \
          acc2[j % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(j < G::CB ? hf0 : hf1, fr[j % PFC], acc2[j % G::CB], 0, 0, 0); \
        if (j + PFC < 2 * G::CB) fr[j % PFC] = *reinterpret_cast<const bf16x8*>(sl + (j + PFC) * 1024);    \
        if (j % DMA_EVERY == 0 && j / DMA_EVERY < NDMA) { BLK_DMA(B, j / DMA_EVERY, ST) }                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
      }                                                                                                    \
    } else {                                                                                               \
      _Pragma("unroll") for (int k = 0; k < NDMA; ++k) { BLK_DMA(B, k, false) }                            \
    }                                                                                                      \
    BLK_SYNC(B, ST, 0)                                                                                     \
  }
  C_BLOCK(0, false)
  C_BLOCK(1, false)
  for (int b = 2; b + 2 < G::NHB; ++b) C_BLOCK(b, true)
  C_BLOCK(G::NHB - 2, false)
  C_BLOCK(G::NHB - 1, false)
  C_BLOCK(G::NHB, false)
  C_BLOCK(G::NHB + 1, false)
#undef C_BLOCK
#undef BLK_DMA
#undef BLK_SYNC
#undef DMA_W1_PIECE
#undef DMA_W2_PIECE
  // ---- epilogue: b2 / gamma through the H buffers (written by the producers), the tile through the dead rings, as blk_mlp_fwd_kernel
  __syncthreads();
  float* scr = reinterpret_cast<float*>(lds) + pair * (16 * C);
#if BLK2_EPI
  // (BLK2_EPI: this wavefront only scatters its accumulators, 16 rows per pass; the producer of the pair - which holds the residual
  //  tile - does the arithmetic and the stores)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        scr[((r & 3) + 8 * (r >> 2) + 4 * half) * C + cb * 32 + l32] = acc2[cb][8 * pass + r];
    __syncthreads();                                                  // scattered: the producer reads
    if (pass == 0) __syncthreads();                                   // ... and is done with these rows
  }
#else
  const float4* b2v = reinterpret_cast<const float4*>(cst);
  const float4* gav = reinterpret_cast<const float4*>(cst + C);
  const TX* resid = static_cast<const TX*>(p.resid);
  TO* out = static_cast<TO*>(p.out);
  constexpr int C4 = C / 4, NCH = 16 * C4 / 64;
  constexpr int GRP = (NCH % 6 == 0 && C < 384) ? 6 : (NCH % 4 == 0 ? 4 : 3);       // residual chunks in flight per lane (256 registers)
  static_assert(NCH % GRP == 0, "chunk groups");
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const long e0 = (m0 + 16 * pass) * C;
    const long e_end = p.M * C;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        scr[((r & 3) + 8 * (r >> 2) + 4 * half) * C + cb * 32 + l32] = acc2[cb][8 * pass + r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g0 = 0; g0 < NCH; g0 += GRP) {
      float4 xv[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int idx = (g0 + j) * 64 + lane;
        const long e = e0 + idx * 4;
        xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (resid && e < e_end) {
          if constexpr (sizeof(TX) == 4) {
            xv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resid) + e);
          } else {
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(resid) + e);
            xv[j] = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int idx = (g0 + j) * 64 + lane;
        const long e = e0 + idx * 4;
        const int c4 = idx % C4;
        const float4 o = reinterpret_cast<const float4*>(scr)[idx];
        const float4 bb = b2v[c4], gg = gav[c4];
        const float y0 = o.x + bb.x, y1 = o.y + bb.y, y2v = o.z + bb.z, y3 = o.w + bb.w;
        if (e < e_end) {
          if (p.y2) *reinterpret_cast<uint2*>(p.y2 + e) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2v, y3));
          const float o0 = fmaf(y0, gg.x, xv[j].x), o1 = fmaf(y1, gg.y, xv[j].y);
          const float o2 = fmaf(y2v, gg.z, xv[j].z), o3 = fmaf(y3, gg.w, xv[j].w);
          if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + e) = make_float4(o0, o1, o2, o3);
          else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + e) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
        }
      }
    }
  }
#endif
}

// Which forward kernel serves width C?  Measured (tools/mlp_bench.py, batch 256, profiles/r05_fused_mlp.md): the wavefront-pair kernel is
// ahead of the single-wavefront one at C = 384 and C = 256 in all three forms (no workspace / Hpre / training outputs).  APGD_BLK2
// sets the start-up value: a list of widths ("" = the single-wavefront kernel everywhere); cnx_runtime_switch(CNX_SWITCH_BLK2_WIDTHS)
// changes it in a running process (bench.py's interleaved A/B leg).  Bit 0: C = 256, bit 1: C = 384.
int& blk2_widths() {
  static int m = [] {
    const char* env = getenv("APGD_BLK2");
    return env ? ((strstr(env, "256") ? 1 : 0) | (strstr(env, "384") ? 2 : 0)) : 3;
  }();
  return m;
}
inline bool use_blk2(int C, bool ws) {
  (void)ws;
  const int m = blk2_widths();
  return (C == 256 && (m & 1)) || (C == 384 && (m & 2));
}

template <int C>
int launch_blk2_fwd(const BlkFwdArgs& a, int resid_dtype, int out_dtype, hipStream_t s) {
  using G = Geo2<C>;
  const dim3 grid(static_cast<unsigned>((a.M + 127) / 128)), block(512);
#define BLK2_LAUNCH_WS(TX, TO, WSV)                                                                              \
  {                                                                                                              \
    auto kfn = blk2_fwd_kernel<C, TX, TO, WSV>;                                                                  \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS); \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, G::LDS, s, a);                                                          \
  }
#define BLK2_LAUNCH(TX, TO) { if (a.hpre) { if (a.hact) BLK2_LAUNCH_WS(TX, TO, 2) else BLK2_LAUNCH_WS(TX, TO, 1) } else BLK2_LAUNCH_WS(TX, TO, 0) }
  if (resid_dtype == APGD_F32 && out_dtype == APGD_F32) BLK2_LAUNCH(float, float)
  else if (resid_dtype == APGD_F32) BLK2_LAUNCH(float, uint16_t)
  else if (out_dtype == APGD_F32) BLK2_LAUNCH(uint16_t, float)
  else BLK2_LAUNCH(uint16_t, uint16_t)
#undef BLK2_LAUNCH
#undef BLK2_LAUNCH_WS
  return launch_status();
}

// Wavefronts per workgroup of blk_mlp_fwd_kernel at width C: 8 (256 rows share one weight stream, round 6) where the registers allow two
// wavefronts per SIMD, else 4.  cnx_runtime_switch(CNX_SWITCH_FWD_WAVES8, mask) / APGD_FWD_W8: bit 0 = C 128, bit 1 = C 192.
int& fwd_w8_widths() {
  static int m = [] {
    const char* env = getenv("APGD_FWD_W8");
    return env ? ((strstr(env, "128") ? 1 : 0) | (strstr(env, "192") ? 2 : 0)) : kFwdW8Default;
  }();
  return m;
}

template <int C, int W>
int launch_blk_fwd_w(const BlkFwdArgs& a, int resid_dtype, int out_dtype, hipStream_t s) {
  using G = Geo<C, W>;
  const dim3 grid(static_cast<unsigned>((a.M + G::BM - 1) / G::BM)), block(64 * W);
#define BLK_LAUNCH(TX, TO) { if (a.hpre) { if constexpr (G::PIPE) { if (a.hact) BLK_LAUNCH_WS(TX, TO, 2) else BLK_LAUNCH_WS(TX, TO, 1) } else return APGD_ERR_ARG; } else BLK_LAUNCH_WS(TX, TO, 0) }
#define BLK_LAUNCH_WS(TX, TO, WSV)                                                                               \
  {                                                                                                              \
    auto kfn = blk_mlp_fwd_kernel<C, TX, TO, WSV, W>;                                                            \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                G::FWD_LDS);                                                                     \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, G::FWD_LDS, s, a);                                                      \
  }
  if (resid_dtype == APGD_F32 && out_dtype == APGD_F32) BLK_LAUNCH(float, float)
  else if (resid_dtype == APGD_F32) BLK_LAUNCH(float, uint16_t)
  else if (out_dtype == APGD_F32) BLK_LAUNCH(uint16_t, float)
  else BLK_LAUNCH(uint16_t, uint16_t)
#undef BLK_LAUNCH
#undef BLK_LAUNCH_WS
  return launch_status();
}

template <int C>
int launch_blk_fwd(const BlkFwdArgs& a, int resid_dtype, int out_dtype, hipStream_t s) {
#if BLK_FWD_W8_BUILD
  if constexpr (C == 128 || C == 192) {
    if (fwd_w8_widths() & (C == 128 ? 1 : 2)) return launch_blk_fwd_w<C, 8>(a, resid_dtype, out_dtype, s);
  }
#endif
  return launch_blk_fwd_w<C, 4>(a, resid_dtype, out_dtype, s);
}


// =====================================================================================================================
// Backward of the block tail w.r.t. the LayerNorm output a = LN(u)  (input gradient of models/convnext.py:41-49):
//     dO = g * gamma                       (g = d loss / d block output)
//     dH = dO W2            Hpre = a W1^T + b1 (recomputed)          dHpre = dH * GELU'(Hpre)
//     da = dHpre W1
// Same decomposition as the forward: a wavefront owns 32 rows; a and dO live in registers as B-operand fragments,
// the 32 x C fp32 da tile in accumulators; per 32-wide hidden slice three MFMA GEMMs
//     GEMM1  Hpre^T[h][m] = W1[slice]    (A, LDS) x a^T  (B, regs)          k = channel
//     GEMM2  dH^T  [h][m] = W2[:,slice]^T (A, LDS) x dO^T (B, regs)          k = channel
//     GEMM3  da[m][c]    += dHpre (A, regs: the accumulator layout again) x W1[slice] (B, LDS)   k = hidden
// When `emit` outputs are given (training backward) the kernel also writes what the weight gradients need:
// a and dO as [M, C] bf16 and H^T = GELU(Hpre)^T, dHpre^T as [4C, M] bf16 (K-contiguous operands for
//     dW1 = dHpre^T a,   dW2^T = H^T dO ).
// Wb: [NHB][3C/16 pieces][64 lanes][8] bf16: pieces [0,KS) = W1 A-fragments (as in the forward pack),
// [KS, 2KS) = W2^T A-fragments, [2KS, 2KS + 2CB) = W1 B-fragments (cb, t).
template <typename TW>
__global__ __launch_bounds__(256) void pack_bwd_kernel(const TW* __restrict__ W1, const TW* __restrict__ W2,
                                                       uint16_t* __restrict__ Wb, int C) {
  const int KS = C / 16, PIECES = 2 * KS + 2 * (C / 32);
  const long total = static_cast<long>(C / 8) * PIECES * 64;
  const long q = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= total) return;
  const int lane = static_cast<int>(q & 63), l32 = lane & 31, half = lane >> 5;
  const int p = static_cast<int>((q >> 6) % PIECES);
  const int hb = static_cast<int>((q >> 6) / PIECES);
  float v[8];
  if (p < KS) {                                   // W1[h][c], lane = h, k = channel
    const TW* src = W1 + static_cast<long>(hb * 32 + l32) * C + half * (C / 2) + p * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = static_cast<float>(src[e]);
  } else if (p < 2 * KS) {                        // W2[c][h], lane = h, k = channel
    const int ks = p - KS;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = static_cast<float>(W2[static_cast<long>(half * (C / 2) + ks * 8 + e) * (4 * C) + hb * 32 + l32]);
  } else {                                        // W1[h][c], lane = c, k = hidden (accumulator order)
    const int cb = (p - 2 * KS) >> 1, t = (p - 2 * KS) & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = static_cast<float>(W1[static_cast<long>(hb * 32 + (e & 3) + 8 * (2 * t + (e >> 2)) + 4 * half) * C + cb * 32 + l32]);
  }
  uint4 o;
  o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
  reinterpret_cast<uint4*>(Wb)[q] = o;
}

struct BlkBwdArgs {
  const uint16_t* u;       // [M, C] bf16 depthwise-conv output
  const float* ln_w;       // [C]
  const float* ln_b;       // [C]
  const float* mean;       // [M]  (saved by the forward)
  const float* rstd;       // [M]
  const void* g;           // [M, C] TG: gradient w.r.t. the block output
  const float* gamma;      // [C] or NULL
  const uint16_t* Wb;      // packed backward weights
  const float* b1;         // [4C]
  uint16_t* da;            // [M, C] bf16: gradient w.r.t. LN(u)  (LNB kernels: w.r.t. u itself)
  uint16_t* a_out;         // emit: [M, C] bf16 LN(u)           (all four NULL or all four set)
  uint16_t* do_out;        // emit: [M, C] bf16 g * gamma
  uint16_t* ht_out;        // emit: [4C, M] bf16 GELU(Hpre)^T
  uint16_t* dhpt_out;      // emit: [4C, M] bf16 dHpre^T
  const uint16_t* hpre;    // HPRE kernels: the forward's Hpre workspace (cnx_block_mlp_fwd_hpre), else unused
  long M;
  long a_stride;           // row stride of a_out in elements (>= C; lets the caller append a ones column for d(b1))
  int emit_acc;            // emit mode 2: ht_out / dhpt_out are CNX_TN_ACC tiles of H / dHpre ([M/32][4C/32] x 2 KiB), not [4C, M]
};

#ifndef BLK_BWD_PIPE
#define BLK_BWD_PIPE 1
#endif
template <int C>
struct GeoB {
  static constexpr int KS = C / 16, CB = C / 32, NHB = C / 8;
#ifndef BLK_BWD96_WAVES
#define BLK_BWD96_WAVES 4
#endif
  static constexpr int WAVES = (C <= 96) ? BLK_BWD96_WAVES : 4;
  static constexpr int PIECES = 2 * KS + 2 * CB;
  static constexpr int SLICE = PIECES * 1024;
  static constexpr int ROUNDS = (PIECES + WAVES - 1) / WAVES;      // DMA instructions per wavefront per slice (upper bound)
  static constexpr int MIN_ROUNDS = PIECES / WAVES;                // ... lower bound (the counted wait must use this one)
  static constexpr int DEPTH = 3;
  static constexpr int LDS = DEPTH * SLICE + 16 * C;                // + b1 (4C fp32)
  static constexpr int LDS_EMIT = LDS + WAVES * 2048;                // + one 32 x 32 bf16 transpose tile per wavefront
  static constexpr int BM = WAVES * 32;
  static_assert(WAVES * 16 * C * 4 <= DEPTH * SLICE, "the epilogue tile reuses the weight ring");
};

// HPRE: Hpre comes from the workspace the pipelined forward wrote (cnx_block_mlp_fwd_hpre) instead of being recomputed - no
// LN(u) operand fragments (C/4 registers less per lane: what makes C = 384 fit one wavefront per SIMD), a third fewer MFMAs,
// and only the W2^T and GEMM3 pieces of a packed slice go through LDS (KS + 2 CB KiB: three ring slots fit at C = 384).
// EMIT: 0 = input gradient only; 1 = also the operands of the weight-gradient GEMMs as rounds 1 - 4 wrote them (a, dO rows; H^T, dHpre^T
// as [4C, M] through an LDS transposition; recomputing kernels only); 2 = a (recomputing kernels) and dO rows, and H (recomputing
// kernels) and dHpre as CNX_TN_ACC tiles - the lane's accumulator-order pairs leave with two 16-byte stores, nothing is transposed:
// cnx_gemm_tn_ex reads that layout (round 5).  With HPRE the forward (WS == 2) has already written H and the LN(u) rows.
template <int C, typename TG, int EMIT, bool LNB, bool HPRE = false>
__global__ __launch_bounds__(GeoB<C>::WAVES * 64, ((C <= 96 || (HPRE && C == 192)) ? 2 : 1)) void blk_mlp_bwd_kernel(const BlkBwdArgs p) {
  using G = GeoB<C>;
  static_assert(!(HPRE && EMIT == 1) && !(EMIT == 1 && LNB), "emit modes: see above");
  // PIPE_R: the recomputing input-gradient kernel at one wavefront per SIMD (C >= 128) runs the software-pipelined loop too
  // (GEMM1 / dH of block t+1 interleaved with the unpacked GELU' of block t); C = 96 (two wavefronts per SIMD, power cap) and the
  // emit mode keep the straight loop.
  constexpr bool PIPE_R = !HPRE && !EMIT && LNB && (C == 128 || C == 192) && BLK_BWD_PIPE;   // (C = 256: the second accumulator set spills)
  constexpr int LP0 = HPRE ? G::KS : 0;                          // first packed piece of a slice that goes through LDS
  constexpr int LPIECES = G::PIECES - LP0, LSLICE = LPIECES * 1024;
  constexpr int LROUNDS = (LPIECES + G::WAVES - 1) / G::WAVES, LMIN_ROUNDS = LPIECES / G::WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ring = lds;
  float* b1s = reinterpret_cast<float*>(lds + G::DEPTH * LSLICE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * G::BM + wave * 32;

  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wb);    // wave-uniform; the lane's part is lane16
  const uint32_t lane16 = lane * 16, ring0 = __builtin_amdgcn_readfirstlane(lds_addr(ring));
#define DMA_SLICE(S)                                                                                       \
  {                                                                                                        \
    const unsigned char* gs = wsrc + static_cast<long>(S) * G::SLICE + LP0 * 1024;                         \
    const uint32_t ls = ring0 + ((S) % G::DEPTH) * LSLICE;                                                 \
    _Pragma("unroll") for (int i = 0; i < LROUNDS; ++i) {                                                  \
      const int piece = i * G::WAVES + wave;                                                               \
      if (piece < LPIECES)                                                                                 \
        glds16(gs + piece * 1024, lane16, ls + piece * 1024); \
    }                                                                                                      \
  }
  if constexpr (!HPRE) {                                  // (the pipelined loops below arrange their ring differently)
    if constexpr (!PIPE_R) {
      DMA_SLICE(0)
      DMA_SLICE(1)
    }
    for (int i = tid; i < C; i += G::WAVES * 64) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  }

  long row = m0 + l32;
  const bool row_ok = row < p.M;
  if (!row_ok) row = p.M - 1;
  // ---- a = LN(u) with the saved statistics, and dO = g * gamma: B-operand fragments (lane = row, k = channel)
  bf16x8 af[HPRE ? 1 : G::KS], gf[G::KS];
  if constexpr (!HPRE) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    const uint4* up = reinterpret_cast<const uint4*>(p.u + row * C + half * (C / 2));
    const float4* lw = reinterpret_cast<const float4*>(p.ln_w + half * (C / 2));
    const float4* lb = reinterpret_cast<const float4*>(p.ln_b + half * (C / 2));
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const uint4 raw = up[ks];
      const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
      const float4 w0 = lw[2 * ks], w1 = lw[2 * ks + 1], c0 = lb[2 * ks], c1 = lb[2 * ks + 1];
      const float gw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      const float o[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      uint32_t pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = fmaf((bf16_lo(w[j]) - mean) * rstd, gw[2 * j], o[2 * j]);
        const float b = fmaf((bf16_hi(w[j]) - mean) * rstd, gw[2 * j + 1], o[2 * j + 1]);
        pk[j] = pack_bf16(a, b);
      }
      const uint4 packed = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      af[ks] = __builtin_bit_cast(bf16x8, packed);
      if (EMIT && row_ok) reinterpret_cast<uint4*>(p.a_out + row * p.a_stride + half * (C / 2))[ks] = packed;
    }
  }
  {
    const float4* gmp = p.gamma ? reinterpret_cast<const float4*>(p.gamma + half * (C / 2)) : nullptr;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      float v[8];
      if constexpr (sizeof(TG) == 4) {
        const float4* gp = reinterpret_cast<const float4*>(static_cast<const float*>(p.g) + row * C + half * (C / 2));
        const float4 g0 = gp[2 * ks], g1 = gp[2 * ks + 1];
        v[0] = g0.x; v[1] = g0.y; v[2] = g0.z; v[3] = g0.w; v[4] = g1.x; v[5] = g1.y; v[6] = g1.z; v[7] = g1.w;
      } else {
        const uint4 raw = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.g) + row * C + half * (C / 2))[ks];
        v[0] = bf16_lo(raw.x); v[1] = bf16_hi(raw.x); v[2] = bf16_lo(raw.y); v[3] = bf16_hi(raw.y);
        v[4] = bf16_lo(raw.z); v[5] = bf16_hi(raw.z); v[6] = bf16_lo(raw.w); v[7] = bf16_hi(raw.w);
      }
      if (gmp) {
        const float4 m0v = gmp[2 * ks], m1v = gmp[2 * ks + 1];
        v[0] *= m0v.x; v[1] *= m0v.y; v[2] *= m0v.z; v[3] *= m0v.w; v[4] *= m1v.x; v[5] *= m1v.y; v[6] *= m1v.z; v[7] *= m1v.w;
      }
      const uint4 packed = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
      gf[ks] = __builtin_bit_cast(bf16x8, packed);
      if (EMIT && row_ok) reinterpret_cast<uint4*>(p.do_out + row * C + half * (C / 2))[ks] = packed;
    }
  }

  f32x16 acc3[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[cb][r] = 0.f;

  if constexpr (HPRE) {
    // ---- software-pipelined hidden loop (one wavefront per SIMD: nothing else hides the activation math).  LDS slice L = t + 1,
    //      t = -1 .. NHB-1, holds [W2^T fragments of block t+1 | GEMM3 fragments of block t]; iteration L:
    //        MFMA stream:  dH(t+1) = dO W2^T (KS)  ->  GEMM3(t) first half (CB, needs pairs 0-3 of dHpre(t))  ->  second half (CB)
    //        VALU stream:  dHpre(t) = dH(t) * GELU'(Hpre(t)), 16 values per lane in UNPACKED instructions behind the first KS + CB MFMAs
    //      Hpre(t+1) is loaded (2 x 16 bytes per lane) at the top of iteration L and converted at the top of L + 1.
    static_assert(LPIECES % G::WAVES == 0 && G::NHB % 2 == 0, "uniform DMA count per slice; two-iteration unroll");
    constexpr int NF = G::KS + 2 * G::CB, PF = 4, SLOTS = G::KS + G::CB, NUOP = 4 * 62, DMA_EVERY = SLOTS / LROUNDS;
    static_assert(DMA_EVERY >= 1 && NUOP * G::KS / SLOTS >= 124, "pairs 0-3 are ready when GEMM3 starts");
    float c6v = 1.8761737253e-03f;                        // leading coefficient of W(x) in a VGPR (one constant-bus operand per VOP3)
    asm volatile("" : "+v"(c6v));
    const long tile = static_cast<long>(blockIdx.x) * G::WAVES + wave;
#define H_DMA_PIECE(L, Q)                                                                                      \
    {                                                                                                          \
      const int q_ = (Q) * G::WAVES + wave;                /* compact piece: < KS W2^T of block L, else GEMM3 of block L-1 */ \
      const int blk_ = q_ < G::KS ? ((L) < G::NHB ? (L) : G::NHB - 1) : ((L) > 0 ? (L) - 1 : 0);   /* (steady iterations: 0 < L < NHB) */ \
      glds16(wsrc + static_cast<long>(blk_) * G::SLICE + (LP0 + q_) * 1024, lane16, ring0 + ((L) % G::DEPTH) * LSLICE + q_ * 1024);   \
    }
#define H_LOAD_HPRE(DST, T)                                                                                    \
    {                                                                                                          \
      const uint4* hp_ = reinterpret_cast<const uint4*>(p.hpre) + (tile * G::NHB + ((T) < G::NHB ? (T) : G::NHB - 1)) * 128 + l32 * 4 + half * 2; \
      DST[0] = hp_[0]; DST[1] = hp_[1];                                                                        \
    }
#pragma unroll
    for (int q = 0; q < LROUNDS; ++q) H_DMA_PIECE(0, q)
#pragma unroll
    for (int q = 0; q < LROUNDS; ++q) H_DMA_PIECE(1, q)
    uint4 hra[2], hrb[2];
    f32x16 dha, dhb;
    {                                                     // L = 0: dH of block 0 only
      H_LOAD_HPRE(hra, 0)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LROUNDS + 2) : "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < LROUNDS; ++q) H_DMA_PIECE(2, q)
      const unsigned char* sl = ring + lane * 16;
      bf16x8 fr[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);
#pragma unroll
      for (int r = 0; r < 16; ++r) dha[r] = 0.f;
#pragma unroll
      for (int i = 0; i < G::KS; ++i) {
        dha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[i], dha, 0, 0, 0);
        if (i + PF < G::KS) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      }
    }
    // ST ("steady"): compile-time promise that slices L + 1 and L + 2 exist - straight-line code, no branch inside the loop body
    // (see blk_mlp_fwd_kernel); the last two iterations are instantiated with a constant L
#define H_ITER(L, HCUR, HNEXT, DHIN, DHOUT, ST)                                                                \
    {                                                                                                          \
      H_LOAD_HPRE(HNEXT, L)                               /* Hpre of block t+1 = L, used by the next iteration */ \
      if (ST || (L) + 1 <= G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LROUNDS + 2) : "memory");          \
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                    \
      __builtin_amdgcn_s_barrier();                                                                            \
      const unsigned char* sl = ring + ((L) % G::DEPTH) * LSLICE + lane * 16;                                  \
      bf16x8 fr[PF];                                                                                           \
      _Pragma("unroll") for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);  \
      float zq[16];                                                                                            \
      {                                                                                                        \
        const uint32_t hw_[8] = {HCUR[0].x, HCUR[0].y, HCUR[0].z, HCUR[0].w, HCUR[1].x, HCUR[1].y, HCUR[1].z, HCUR[1].w}; \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) { zq[2 * k] = bf16_lo(hw_[k]); zq[2 * k + 1] = bf16_hi(hw_[k]); } \
      }                                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) DHOUT[r] = 0.f;                                           \
      float gx[4], ge[4], gw[4];                                                                               \
      uint32_t pk[8];                                                                                          \
      bf16x8 dhf0, dhf1;                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int i = 0; i < SLOTS; ++i) {                                                      \
        if (i == G::KS) dhf0 = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));             \
        if (i < G::KS) DHOUT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[i < G::KS ? i : 0], DHOUT, 0, 0, 0); \
        else acc3[(i - G::KS) % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dhf0, fr[i % PF], acc3[(i - G::KS) % G::CB], 0, 0, 0); \
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + gemm3_piece(i + PF) * 1024);       \
        if (i % DMA_EVERY == 0 && i / DMA_EVERY < LROUNDS && (ST || (L) + 2 <= G::NHB)) H_DMA_PIECE((L) + 2, i / DMA_EVERY) \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int uo = NUOP * i / SLOTS; uo < NUOP * (i + 1) / SLOTS; ++uo) {                 \
          const int qd = uo / 62;                                                                              \
          const float z4[4] = {zq[4 * qd], zq[4 * qd + 1], zq[4 * qd + 2], zq[4 * qd + 3]};                    \
          const float d4[4] = {DHIN[4 * qd], DHIN[4 * qd + 1], DHIN[4 * qd + 2], DHIN[4 * qd + 3]};            \
          gelu_grad_uop(uo % 62, z4, d4, gx, ge, gw, pk[2 * qd], pk[2 * qd + 1], c6v);                         \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      }                                                                                                        \
      dhf1 = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));                               \
      if constexpr (EMIT == 2) {   /* dHpre of block t = L - 1 in its Hpre's tile (accumulator order: CNX_TN_ACC) */ \
        uint4* dd_ = reinterpret_cast<uint4*>(p.dhpt_out) + (tile * G::NHB + ((L) - 1)) * 128 + l32 * 4 + half * 2; \
        dd_[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);                                                       \
        dd_[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);                                                       \
      }                                                                                                        \
      _Pragma("unroll") for (int j = G::CB; j < 2 * G::CB; ++j) {                                              \
        const int i = G::KS + j;                                                                               \
        acc3[j - G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dhf1, fr[i % PF], acc3[j - G::CB], 0, 0, 0); \
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + gemm3_piece(i + PF) * 1024);       \
      }                                                                                                        \
    }
    // fragment i of an iteration's stream -> piece of the compact LDS slice: the KS W2^T pieces in order, then GEMM3's in
    // (t, cb) order (consecutive MFMAs update different accumulators) out of the packed (cb, t) order
    auto gemm3_piece = [](int i) constexpr {
      if (i < G::KS) return i;
      const int j = i - G::KS;
      return G::KS + (j % G::CB) * 2 + (j / G::CB);
    };
    static_assert(G::NHB >= 4, "steady iterations 1 .. NHB-2, then the constant-L tail");
    for (int L = 1; L + 1 <= G::NHB - 2; L += 2) {
      H_ITER(L, hra, hrb, dha, dhb, true)
      H_ITER(L + 1, hrb, hra, dhb, dha, true)
    }
    H_ITER(G::NHB - 1, hra, hrb, dha, dhb, false)
    H_ITER(G::NHB, hrb, hra, dhb, dha, false)
#undef H_ITER
#undef H_LOAD_HPRE
#undef H_DMA_PIECE
  }
  if constexpr (PIPE_R) {
    // ---- software-pipelined recomputing loop.  LDS slice L = t + 1 holds [W1(t+1) | W2^T(t+1) | GEMM3 pieces of block t]; iteration L:
    //        MFMA stream:  Hpre(t+1) = a W1^T + b1 and dH(t+1) = dO W2^T, alternating (2 KS)  ->  GEMM3(t) (CB + CB)
    //        VALU stream:  dHpre(t) = dH(t) * GELU'(Hpre(t)) in unpacked instructions behind the first 2 KS + CB MFMAs
    static_assert(G::PIECES % G::WAVES == 0 && G::NHB % 2 == 0, "uniform DMA count per slice; two-iteration unroll");
    constexpr int NG = 2 * G::KS, NF = NG + 2 * G::CB, PF = 4, SLOTS = NG + G::CB, NUOP = 4 * 62, RND = G::PIECES / G::WAVES;
    constexpr int DMA_EVERY = SLOTS / RND;
    static_assert(DMA_EVERY >= 1 && NUOP * NG / SLOTS >= 124, "pairs 0-3 are ready when GEMM3 starts");
    float c6v = 1.8761737253e-03f;
    asm volatile("" : "+v"(c6v));
#define R_DMA_PIECE(L, Q)                                                                                      \
    {                                                                                                          \
      const int q_ = (Q) * G::WAVES + wave;                /* piece < 2 KS: W1 / W2^T of block L, else GEMM3 of block L-1 */ \
      const int blk_ = q_ < NG ? ((L) < G::NHB ? (L) : G::NHB - 1) : ((L) > 0 ? (L) - 1 : 0);                  \
      glds16(wsrc + static_cast<long>(blk_) * G::SLICE + q_ * 1024, lane16, ring0 + ((L) % G::DEPTH) * G::SLICE + q_ * 1024); \
    }
#define R_BIAS(Z, T)                                                                                           \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                         \
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + ((T) < G::NHB ? (T) : G::NHB - 1) * 32 + 8 * g4 + 4 * half); \
      Z[4 * g4 + 0] = b4.x; Z[4 * g4 + 1] = b4.y; Z[4 * g4 + 2] = b4.z; Z[4 * g4 + 3] = b4.w;                  \
    }
    auto r_piece = [](int i) constexpr {                  // fragment i of an iteration's stream -> piece of the slice
      if (i < NG) return (i & 1) ? G::KS + (i >> 1) : (i >> 1);
      const int j = i - NG;                               // j = t * CB + cb  ->  packed piece (cb, t)
      return NG + (j % G::CB) * 2 + (j / G::CB);
    };
#pragma unroll
    for (int q = 0; q < RND; ++q) R_DMA_PIECE(0, q)
#pragma unroll
    for (int q = 0; q < RND; ++q) R_DMA_PIECE(1, q)
    f32x16 za, zb, dha, dhb;
    {                                                     // L = 0: Hpre and dH of block 0 only
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RND) : "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int q = 0; q < RND; ++q) R_DMA_PIECE(2, q)
      const unsigned char* sl = ring + lane * 16;
      bf16x8 fr[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + r_piece(i) * 1024);
      R_BIAS(za, 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) dha[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        if (i & 1) dha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[i >> 1], dha, 0, 0, 0);
        else za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i >> 1], za, 0, 0, 0);
        if (i + PF < NG) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + r_piece(i + PF) * 1024);
      }
    }
#define R_ITER(L, ZIN, DHIN, ZOUT, DHOUT, ST)                                                                  \
    {                                                                                                          \
      if (ST || (L) + 1 <= G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RND) : "memory");                  \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
      __builtin_amdgcn_s_barrier();                                                                            \
      const unsigned char* sl = ring + ((L) % G::DEPTH) * G::SLICE + lane * 16;                                \
      bf16x8 fr[PF];                                                                                           \
      _Pragma("unroll") for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + r_piece(i) * 1024); \
      R_BIAS(ZOUT, L)                                                                                          \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) DHOUT[r] = 0.f;                                           \
      float gx[4], ge[4], gw[4];                                                                               \
      uint32_t pk[8];                                                                                          \
      bf16x8 dhf0, dhf1;                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int i = 0; i < SLOTS; ++i) {                                                      \
        if (i == NG) dhf0 = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));                \
        if (i < NG && (i & 1)) DHOUT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[(i < NG ? i : 0) >> 1], DHOUT, 0, 0, 0); \
        else if (i < NG) ZOUT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[(i < NG ? i : 0) >> 1], ZOUT, 0, 0, 0); \
        else acc3[(i - NG) % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dhf0, fr[i % PF], acc3[(i - NG) % G::CB], 0, 0, 0); \
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + r_piece(i + PF) * 1024);           \
        if (i % DMA_EVERY == 0 && i / DMA_EVERY < RND && (ST || (L) + 2 <= G::NHB)) R_DMA_PIECE((L) + 2, i / DMA_EVERY) \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int uo = NUOP * i / SLOTS; uo < NUOP * (i + 1) / SLOTS; ++uo) {                 \
          const int qd = uo / 62;                                                                              \
          const float z4[4] = {ZIN[4 * qd], ZIN[4 * qd + 1], ZIN[4 * qd + 2], ZIN[4 * qd + 3]};                \
          const float d4[4] = {DHIN[4 * qd], DHIN[4 * qd + 1], DHIN[4 * qd + 2], DHIN[4 * qd + 3]};            \
          gelu_grad_uop(uo % 62, z4, d4, gx, ge, gw, pk[2 * qd], pk[2 * qd + 1], c6v);                         \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      }                                                                                                        \
      dhf1 = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));                               \
      _Pragma("unroll") for (int j = G::CB; j < 2 * G::CB; ++j) {                                              \
        const int i = NG + j;                                                                                  \
        acc3[j - G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dhf1, fr[i % PF], acc3[j - G::CB], 0, 0, 0); \
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + r_piece(i + PF) * 1024);           \
      }                                                                                                        \
    }
    for (int L = 1; L + 1 <= G::NHB - 2; L += 2) {
      R_ITER(L, za, dha, zb, dhb, true)
      R_ITER(L + 1, zb, dhb, za, dha, true)
    }
    R_ITER(G::NHB - 1, za, dha, zb, dhb, false)
    R_ITER(G::NHB, zb, dhb, za, dha, false)
#undef R_ITER
#undef R_BIAS
#undef R_DMA_PIECE
  }
  for (int s = 0; s < ((HPRE || PIPE_R) ? 0 : G::NHB); ++s) {        // the straight recomputing loop (C = 96, emit mode)
    if (s + 1 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LMIN_ROUNDS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (the DMA of slice s + 2 is issued one instruction at a time between the MFMAs below: a burst here stalls the in-order
    //  wavefront at issue while the texture path drains - measured on the forward, profiles/r02_power_and_overlap.md)
    const unsigned char* sl = ring + (s % G::DEPTH) * LSLICE + lane * 16;

    f32x16 acc1, acc2;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + s * 32 + 8 * g4 + 4 * half);
      acc1[4 * g4 + 0] = b4.x; acc1[4 * g4 + 1] = b4.y; acc1[4 * g4 + 2] = b4.z; acc1[4 * g4 + 3] = b4.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    // one stream of NF = 2 KS + 2 CB operand fragments per slice, read PF fragments ahead of the MFMA that consumes them
    // (as in the forward): fragment i < 2 KS alternates W1 / W2^T k-steps (two independent accumulation chains), then the
    // W1 B-fragments of GEMM3 in (t, cb) order so that consecutive MFMAs update different accumulators
    constexpr int NG = 2 * G::KS;                             // MFMAs before the activation
    constexpr int NF = NG + 2 * G::CB, PF = 4;
    auto piece_of = [](int i) constexpr {
      if (i < NG) return (i & 1) ? G::KS + (i >> 1) : (i >> 1);
      const int j = i - NG;                                   // j = t * CB + cb  ->  packed piece (cb, t)
      return 2 * G::KS + (j % G::CB) * 2 + (j / G::CB);
    };
    bf16x8 fr[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + piece_of(i) * 1024);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (i & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[i >> 1], acc2, 0, 0, 0);
      else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[i >> 1], acc1, 0, 0, 0);
      if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + piece_of(i + PF) * 1024);
      constexpr int DMA_EVERY = NG / LROUNDS;
      static_assert(HPRE || (DMA_EVERY >= 1 && DMA_EVERY * (LROUNDS - 1) < NG), "one DMA instruction per DMA_EVERY MFMAs");
      if (i % DMA_EVERY == 0 && i / DMA_EVERY < LROUNDS && s + 2 < G::NHB) {
        const int piece = (i / DMA_EVERY) * G::WAVES + wave;
        if (piece < LPIECES)
          glds16(wsrc + static_cast<long>(s + 2) * G::SLICE + piece * 1024, lane16, ring0 + ((s + 2) % G::DEPTH) * LSLICE + piece * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 dhf[2];
    {
      uint32_t pk[8];
      uint32_t hk[8];                                     // emit only: GELU(Hpre) pairs
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 E;
        const f32x2 gp = gelu_grad2(acc1[r], acc1[r + 1], E);
        const f32x2 d = (f32x2){acc2[r], acc2[r + 1]} * gp;
        pk[r >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, bf16x2));
        if constexpr (EMIT)
          hk[r >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(gelu_from_grad2(acc1[r], acc1[r + 1], gp, E), bf16x2));
      }
      dhf[0] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      dhf[1] = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
      if constexpr (EMIT == 2) {
        if (m0 < p.M) {                                   // (wave-uniform; M is a multiple of 32 on this path: whole tiles)
          const long tq = ((m0 >> 5) * G::NHB + s) * 128 + l32 * 4 + half * 2;
          uint4* hd = reinterpret_cast<uint4*>(p.ht_out) + tq;
          uint4* dd = reinterpret_cast<uint4*>(p.dhpt_out) + tq;
          hd[0] = make_uint4(hk[0], hk[1], hk[2], hk[3]); hd[1] = make_uint4(hk[4], hk[5], hk[6], hk[7]);
          dd[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]); dd[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
      } else if constexpr (EMIT == 1) {
        if ((p.M & 7) == 0) {
          // [4C, M] operands of the weight-gradient GEMMs: a lane holds 16 hidden units of ONE row, the tensors are
          // contiguous along rows.  2x2 exchange with the neighbouring lane (row m^1) turns the (h, h+1) pairs into
          // (m, m+1) pairs, the 32 x 32 tile goes through 2 KiB of LDS and leaves as 16 bytes (8 rows of one hidden
          // unit) per lane: 2 stores per tile instead of 16 two-byte ones.
          uint32_t* tsc = reinterpret_cast<uint32_t*>(b1s + 4 * C) + wave * 512;
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const uint32_t own = which ? pk[q] : hk[q];
              const uint32_t nbr = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(own), 0xB1, 0xf, 0xf, true));
              // even lane: (own.lo, nbr.lo) -> hidden r = 2q;  odd lane: (nbr.hi, own.hi) -> hidden r = 2q + 1
              const uint32_t v = (lane & 1) ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
              const int r = 2 * q + (lane & 1);
              const int hl = (r & 3) + 8 * (r >> 2) + 4 * half;
              tsc[hl * 16 + (l32 >> 1)] = v;
            }
            __builtin_amdgcn_wave_barrier();
            uint16_t* dst = which ? p.dhpt_out : p.ht_out;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const uint4 v = reinterpret_cast<const uint4*>(tsc)[k * 64 + lane];
              const int hl = k * 16 + (lane >> 2);
              const long m = m0 + (lane & 3) * 8;
              if (m < p.M) *reinterpret_cast<uint4*>(dst + (static_cast<long>(s) * 32 + hl) * p.M + m) = v;
            }
          }
        } else if (row_ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long h = s * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const uint32_t d = pk[r >> 1];
            p.dhpt_out[h * p.M + row] = static_cast<uint16_t>((r & 1) ? (d >> 16) : (d & 0xffffu));
            const uint32_t hv = hk[r >> 1];
            p.ht_out[h * p.M + row] = static_cast<uint16_t>((r & 1) ? (hv >> 16) : (hv & 0xffffu));
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * G::CB; ++j) {
      const int i = NG + j;
      // (pinning the da accumulators in AGPRs with inline-asm MFMAs was measured and dropped: the allocator then parks the a / dO
      //  operand fragments in AGPRs instead - 418 vs 408 us at C = 192)
      acc3[j % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dhf[j / G::CB], fr[i % PF], acc3[j % G::CB], 0, 0, 0);
      if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + piece_of(i + PF) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  }
#undef DMA_SLICE

  // ---- epilogue: acc3[cb][r] = da[m0 + (r&3) + 8*(r>>2) + 4*half][cb*32 + l32]; as in the forward the tile leaves through
  //      the dead weight ring, 16 rows per pass, so that a lane stores 16 bytes (8 bf16) of a contiguous run instead of 2
  __syncthreads();
  // the lane's coordinates are recomputed here from an opaque lane id: kept live across the hidden loop they cost the C = 384
  // kernel (all 512 registers in use) a spilled register
  int lane_e = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  asm volatile("" : "+v"(lane_e));
  const int l32e = lane_e & 31, halfe = lane_e >> 5;
  if constexpr (LNB) {
    // ---- ... and the LayerNorm backward rides along (input-gradient-only calls): with t = ln_w * da and
    //      xh = (u - mean) * rstd,   du = rstd * (t - mean_c(t) - xh * mean_c(t * xh)).
    //      16 rows per pass, 4 lanes per row (lane = row*4 + q; q takes the 8-channel chunks q, q+4, ...): the row sums
    //      are two quad exchanges, a row's four lanes store 64 contiguous bytes per chunk step.
    constexpr int CP = C + 4;                                             // padded row: the 4 lanes x 16 rows spread over the banks
    static_assert(G::WAVES * 16 * CP * 4 <= G::DEPTH * LSLICE, "the epilogue tile reuses the weight ring");
    float* scr = reinterpret_cast<float*>(ring) + wave * (16 * CP);
    constexpr int NJ = C / 32;                                            // 8-channel chunks per lane
    const int rl = lane_e >> 2, q = lane_e & 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          scr[((r & 3) + 8 * (r >> 2) + 4 * halfe) * CP + cb * 32 + l32e] = acc3[cb][8 * pass + r];
      __builtin_amdgcn_wave_barrier();
      const long m = m0 + 16 * pass + rl;
      const long mc = m < p.M ? m : p.M - 1;
      const float mean = p.mean[mc], rstd = p.rstd[mc];
      uint4 ur[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) ur[j] = *reinterpret_cast<const uint4*>(p.u + mc * C + (q + 4 * j) * 8);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = (q + 4 * j) * 8;
        const float4 d0 = *reinterpret_cast<const float4*>(scr + rl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + rl * CP + c0 + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const uint32_t uw[4] = {ur[j].x, ur[j].y, ur[j].z, ur[j].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
          const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
          s1 += t;
          s2 = fmaf(t, xh, s2);
        }
      }
      s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
      s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
      s1 *= (1.0f / C); s2 *= (1.0f / C);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = (q + 4 * j) * 8;
        const float4 d0 = *reinterpret_cast<const float4*>(scr + rl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + rl * CP + c0 + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const uint32_t uw[4] = {ur[j].x, ur[j].y, ur[j].z, ur[j].w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
          const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
          o[e] = rstd * (t - s1 - xh * s2);
        }
        if (m < p.M)
          *reinterpret_cast<uint4*>(p.da + m * C + c0) = make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
      }
    }
  } else {
    float* scr = reinterpret_cast<float*>(ring) + wave * (16 * C);
    constexpr int NCH = 16 * C / 8 / 64;                                  // 8-element chunks per lane and pass (C / 32)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const long e0 = (m0 + 16 * pass) * C;
      const long e_end = p.M * C;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          scr[((r & 3) + 8 * (r >> 2) + 4 * halfe) * C + cb * 32 + l32e] = acc3[cb][8 * pass + r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int idx = j * 64 + lane_e;
        const long e = e0 + idx * 8;
        const float4 lo = reinterpret_cast<const float4*>(scr)[2 * idx], hi = reinterpret_cast<const float4*>(scr)[2 * idx + 1];
        if (e < e_end)
          *reinterpret_cast<uint4*>(p.da + e) = make_uint4(pack_bf16(lo.x, lo.y), pack_bf16(lo.z, lo.w), pack_bf16(hi.x, hi.y), pack_bf16(hi.z, hi.w));
      }
    }
  }
}

template <int C>
int launch_blk_bwd(const BlkBwdArgs& a, int g_dtype, bool ln_bwd, hipStream_t s) {
  using G = GeoB<C>;
  const dim3 grid(static_cast<unsigned>((a.M + G::BM - 1) / G::BM)), block(G::WAVES * 64);
  const int emit = a.a_out != nullptr ? (a.emit_acc ? 2 : 1) : 0;
#define BLK_LAUNCH(TG, EM, LN)                                                                                   \
  {                                                                                                              \
    auto kfn = blk_mlp_bwd_kernel<C, TG, EM, LN>;                                                                \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (EM) == 1 ? G::LDS_EMIT : G::LDS);                                                    \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, (EM) == 1 ? G::LDS_EMIT : G::LDS, s, a);                                     \
  }
  if (g_dtype == APGD_F32) {
    if (emit == 2 && ln_bwd) BLK_LAUNCH(float, 2, true) else if (emit == 2) BLK_LAUNCH(float, 2, false) else if (emit) BLK_LAUNCH(float, 1, false) else if (ln_bwd) BLK_LAUNCH(float, 0, true) else BLK_LAUNCH(float, 0, false)
  } else {
    if (emit == 2 && ln_bwd) BLK_LAUNCH(uint16_t, 2, true) else if (emit == 2) BLK_LAUNCH(uint16_t, 2, false) else if (emit) BLK_LAUNCH(uint16_t, 1, false) else if (ln_bwd) BLK_LAUNCH(uint16_t, 0, true) else BLK_LAUNCH(uint16_t, 0, false)
  }
#undef BLK_LAUNCH
  return launch_status();
}

// =====================================================================================================================
// The Hpre backward on wavefront PAIRS (round 6): blk_mlp_bwd_kernel<C, TG, EMIT, LNB, HPRE = true> with the chain of a row tile split
// as blk2_fwd_kernel splits the forward's -
//   producer (wavefronts 0-3)  dO = bf16(g gamma) rows in registers; per hidden block b: dH(b)^T = W2[:, b]^T x dO^T (KS MFMAs, one
//                              accumulator), the block's Hpre tile from the forward's workspace by LDS-DMA (2 KiB as it lies in memory, two
//                              instructions counted by hand next to the weight pieces - a compiler-visible load would make the compiler
//                              wait for every weight piece in flight, and an inline-asm load into registers cannot be waited for without
//                              the compiler reading those registers first), dHpre(b - 1) = dH(b - 1) * GELU'(Hpre(b - 1)) in unpacked VALU instructions
//                              between those MFMAs, dHpre(b - 1) as bf16 operand pairs -> 2 KiB of LDS;
//   consumer (wavefronts 4-7)  da += dHpre(b - 2) x W1[b - 2] (2 CB MFMAs into the 32 x C fp32 tile), the dHpre tile to the workspace
//                              (EMIT == 2), epilogue: LayerNorm backward (LNB) or the plain da rows.
// Rings as blk2_fwd_kernel: W2^T pieces three slots (two blocks ahead), GEMM3 pieces two slots (one block ahead), both straight out of
// the packed backward slices (cnx_mlp_pack_weights_bwd: [W1 | W2^T | GEMM3] per hidden block; the W1 pieces are not read).  Results are
// bit-identical to the single-wavefront kernel's (same MFMA order per accumulator, same activation arithmetic).
template <int C, typename TG, int EMIT, bool LNB>
__global__ __launch_bounds__(512, 2) void blk2_bwd_kernel(const BlkBwdArgs p) {
  using G = Geo2<C>;
  static_assert(EMIT == 0 || EMIT == 2, "emit modes of the Hpre backward");
  constexpr int SLICE_B = (2 * G::KS + 2 * G::CB) * 1024;             // a packed backward slice: [W1 (KS) | W2^T (KS) | GEMM3 (2 CB)] KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* hbuf = lds + G::W1_RING + G::W2_RING;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3;
  const int l32 = lane & 31, half = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * 128 + pair * 32;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wb);
  const uint32_t lane16 = lane * 16, ring1 = __builtin_amdgcn_readfirstlane(lds_addr(lds)), ring2 = ring1 + G::W1_RING;
  // A pieces: W2^T of block T -> ring slot T % 3;  B pieces: the GEMM3 fragments of block T -> ring slot T % 2
#define DMA_A_PIECE(T, I)                                                                                  \
  {                                                                                                        \
    const int q_ = (I) * 8 + wave;                                                                         \
    glds16(wsrc + static_cast<long>(T) * SLICE_B + (G::KS + q_) * 1024, lane16, ring1 + ((T) % 3) * (G::KS * 1024) + q_ * 1024); \
  }
#define DMA_B_PIECE(T, I)                                                                                  \
  {                                                                                                        \
    const int q_ = (I) * 8 + wave;                                                                         \
    glds16(wsrc + static_cast<long>(T) * SLICE_B + (2 * G::KS + q_) * 1024, lane16, ring2 + ((T) % 2) * (2 * G::CB * 1024) + q_ * 1024); \
  }
  constexpr int R1W = G::KS / 8, R2W = 2 * G::CB / 8, NDMA = R1W + R2W;
  // block B, DMA instruction K of this wavefront: first its GEMM3 pieces of block B - 1 (read in block B + 1), then its W2^T pieces of
  // block B + 2 (read in block B + 2)
#define BLK_DMA(B, K, ST)                                                                                  \
  if ((K) < R2W) { if (ST || ((B) >= 1 && (B) - 1 < G::NHB)) DMA_B_PIECE((B) - 1, (K)) }                   \
  else { if (ST || (B) + 2 < G::NHB) DMA_A_PIECE((B) + 2, (K) - R2W) }
  const long tile = static_cast<long>(blockIdx.x) * 4 + pair;
  unsigned char* hb_lane = hbuf + pair * 8192 + lane * 32;           // buffer i at + 4096 i: the dHpre tile of a block
  // the Hpre tile of block b ((tile NHB + b) 2048 bytes into the workspace) goes, as it lies, into the second half of buffer b & 1: issued
  // by the CONSUMER at the top of its block b (it has the lighter instruction stream), read by the producer at the top of block b + 1
  const unsigned char* hp_base = reinterpret_cast<const unsigned char*>(p.hpre) + tile * G::NHB * 2048;
  const uint32_t hp_lds = __builtin_amdgcn_readfirstlane(lds_addr(hbuf)) + pair * 8192 + 2048;

  if (wave < 4) {
    // ================================================================ producer: dO rows, dH, GELU', dHpre -> LDS
    if constexpr (BLK2_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    long row = m0 + l32;
    const bool row_ok = row < p.M;
    if (!row_ok) row = p.M - 1;
    bf16x8 gf[G::KS];
    {
      const float4* gmp = p.gamma ? reinterpret_cast<const float4*>(p.gamma + half * (C / 2)) : nullptr;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        float v[8];
        if constexpr (sizeof(TG) == 4) {
          const float4* gp = reinterpret_cast<const float4*>(static_cast<const float*>(p.g) + row * C + half * (C / 2));
          const float4 g0 = gp[2 * ks], g1 = gp[2 * ks + 1];
          v[0] = g0.x; v[1] = g0.y; v[2] = g0.z; v[3] = g0.w; v[4] = g1.x; v[5] = g1.y; v[6] = g1.z; v[7] = g1.w;
        } else {
          const uint4 raw = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.g) + row * C + half * (C / 2))[ks];
          v[0] = bf16_lo(raw.x); v[1] = bf16_hi(raw.x); v[2] = bf16_lo(raw.y); v[3] = bf16_hi(raw.y);
          v[4] = bf16_lo(raw.z); v[5] = bf16_hi(raw.z); v[6] = bf16_lo(raw.w); v[7] = bf16_hi(raw.w);
        }
        if (gmp) {
          const float4 m0v = gmp[2 * ks], m1v = gmp[2 * ks + 1];
          v[0] *= m0v.x; v[1] *= m0v.y; v[2] *= m0v.z; v[3] *= m0v.w; v[4] *= m1v.x; v[5] *= m1v.y; v[6] *= m1v.z; v[7] *= m1v.w;
        }
        const uint4 packed = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        gf[ks] = __builtin_bit_cast(bf16x8, packed);
        if (EMIT && row_ok) reinterpret_cast<uint4*>(p.do_out + row * C + half * (C / 2))[ks] = packed;
      }
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(gf[ks]));   // the rows are in registers before the DMA is issued
#pragma unroll
    for (int i = 0; i < R1W; ++i) DMA_A_PIECE(0, i)
#pragma unroll
    for (int i = 0; i < R1W; ++i) DMA_A_PIECE(1, i)
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // W2^T(0), W2^T(1): this wavefront's pieces (and its dO row stores)
    __syncthreads();                                                  // ... everybody's
    float c6v = 1.8761737253e-03f;
    asm volatile("" : "+v"(c6v));
    // the lane's 16 values of a Hpre tile: + 64 l32 + 32 half of the copy in the hand-over buffer
    const unsigned char* hp_lane = hbuf + pair * 8192 + 2048 + l32 * 64 + half * 32;
    // Block b:  MFMA stream  dH(b) = W2[:, b]^T x dO^T (KS MFMAs, b < NHB); the Hpre(b) tile is requested at the top
    //           VALU stream  dHpre(b - 1) = dH(b - 1) * GELU'(Hpre(b - 1)) between those MFMAs (b >= 1) -> hand-over buffer
    constexpr int PF = 4, NUOP = 4 * 62;
    constexpr int P_DMA_EVERY = G::KS / NDMA;
    static_assert(P_DMA_EVERY >= 1 && P_DMA_EVERY * NDMA <= G::KS, "one DMA instruction per P_DMA_EVERY MFMAs");
    // the end of a block: everything but this block's W2^T pieces is in; barrier
#define PB_SYNC(B, ST)                                                                                     \
    if (ST || (B) + 2 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R1W) : "memory");                 \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_s_barrier();
#define PB_BLOCK(B, DCUR, DPREV, ST)                                                                       \
    {                                                                                                      \
      uint32_t pk[8];                                                                                      \
      float gx[4], ge[4], gw[4], zq[16];                                                                   \
      if (ST || (B) >= 1) {                                                                                \
        const uint4* hr_ = reinterpret_cast<const uint4*>(hp_lane + (((B) - 1) & 1) * 4096);               \
        const uint4 h0_ = hr_[0], h1_ = hr_[1];                                                            \
        const uint32_t hw_[8] = {h0_.x, h0_.y, h0_.z, h0_.w, h1_.x, h1_.y, h1_.z, h1_.w};                  \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) { zq[2 * k] = bf16_lo(hw_[k]); zq[2 * k + 1] = bf16_hi(hw_[k]); } \
      }                                                                                                    \
      if (ST || (B) < G::NHB) {                                                                            \
        const unsigned char* sl = lds + ((B) % 3) * (G::KS * 1024) + lane * 16;                            \
        bf16x8 fr[PF];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024); \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) DCUR[r] = 0.f;                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int i = 0; i < G::KS; ++i) {                                                \
          DCUR = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], gf[i], DCUR, 0, 0, 0);                \
          if (i + PF < G::KS) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);         \
          if (i % P_DMA_EVERY == 0 && i / P_DMA_EVERY < NDMA) { BLK_DMA(B, i / P_DMA_EVERY, ST) }          \
          __builtin_amdgcn_sched_barrier(0);                                                               \
          if (ST || (B) >= 1) {                                                                            \
            _Pragma("unroll") for (int uo = NUOP * i / G::KS; uo < NUOP * (i + 1) / G::KS; ++uo) {         \
              const int qd = uo / 62;                                                                      \
              const float z4[4] = {zq[4 * qd], zq[4 * qd + 1], zq[4 * qd + 2], zq[4 * qd + 3]};            \
              const float d4[4] = {DPREV[4 * qd], DPREV[4 * qd + 1], DPREV[4 * qd + 2], DPREV[4 * qd + 3]}; \
              gelu_grad_uop(uo % 62, z4, d4, gx, ge, gw, pk[2 * qd], pk[2 * qd + 1], c6v);                 \
            }                                                                                              \
          }                                                                                                \
          __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                  \
        /* the chain's result is read by inline-asm VALU instructions in the next block: the wait states by hand (blk2_fwd_kernel) */ \
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(DCUR));                                                  \
      } else {                                                                                             \
        _Pragma("unroll") for (int k_ = 0; k_ < NDMA; ++k_) { BLK_DMA(B, k_, false) }                      \
        if ((B) >= 1) {                                                                                    \
          _Pragma("unroll") for (int qd = 0; qd < 4; ++qd) {                                               \
            const float z4[4] = {zq[4 * qd], zq[4 * qd + 1], zq[4 * qd + 2], zq[4 * qd + 3]};              \
            const float d4[4] = {DPREV[4 * qd], DPREV[4 * qd + 1], DPREV[4 * qd + 2], DPREV[4 * qd + 3]};  \
            _Pragma("unroll") for (int uo = 0; uo < 62; ++uo) gelu_grad_uop(uo, z4, d4, gx, ge, gw, pk[2 * qd], pk[2 * qd + 1], c6v); \
          }                                                                                                \
        }                                                                                                  \
      }                                                                                                    \
      if (ST || (B) >= 1) {                                                                                \
        uint4* hw = reinterpret_cast<uint4*>(hb_lane + (((B) - 1) & 1) * 4096);                            \
        hw[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);                                                    \
        hw[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);                                                    \
      }                                                                                                    \
      PB_SYNC(B, ST)                                                                                       \
    }
    static_assert(G::NHB % 2 == 0 && G::NHB >= 6, "two-block unroll, steady blocks 1 .. NHB - 3");
    f32x16 da_, db_;
    PB_BLOCK(0, da_, db_, false)
    for (int b = 1; b + 1 < G::NHB - 2; b += 2) {                     // blocks 1 .. NHB - 4 (pairs), all conditions true
      PB_BLOCK(b, db_, da_, true)
      PB_BLOCK(b + 1, da_, db_, true)
    }
    PB_BLOCK(G::NHB - 3, db_, da_, true)
    PB_BLOCK(G::NHB - 2, da_, db_, false)
    PB_BLOCK(G::NHB - 1, db_, da_, false)
    PB_BLOCK(G::NHB, da_, db_, false)
#if BLK2_EPI
    // (as blk2_fwd_kernel) this wavefront is idle from here on: it fetches what the epilogue reads from memory - the pair's u rows and
    // LayerNorm statistics - under the consumer's last GEMM3 block, and then does the epilogue's arithmetic and stores; the consumer only
    // scatters its accumulators
    constexpr int ENJ = C / 32;                                         // 8-channel chunks per lane (4 lanes per row, 16 rows per pass)
    const int erl = lane >> 2, eq = lane & 3;
    uint4 eur[LNB ? 2 : 1][LNB ? ENJ : 1];
    float emean[2], erstd[2];
    if constexpr (LNB) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const long m = m0 + 16 * pass + erl;
        const long mc = m < p.M ? m : p.M - 1;
        emean[pass] = p.mean[mc]; erstd[pass] = p.rstd[mc];
#pragma unroll
        for (int j = 0; j < ENJ; ++j) eur[pass][j] = *reinterpret_cast<const uint4*>(p.u + mc * C + (eq + 4 * j) * 8);
      }
    }
#endif
    __builtin_amdgcn_s_barrier();                                     // block NHB + 1: the consumers' last GEMM3
#undef PB_BLOCK
#undef PB_SYNC
    __syncthreads();                                                  // the consumers' rings are dead: their epilogue may begin
#if BLK2_EPI
    if constexpr (LNB) {
      constexpr int CP = C + 4;
      const float* scr = reinterpret_cast<const float*>(lds) + pair * (16 * CP);
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                              // the consumer has scattered this pass's 16 rows
        const long m = m0 + 16 * pass + erl;
        const float mean = emean[pass], rstd = erstd[pass];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < ENJ; ++j) {
          const int c0 = (eq + 4 * j) * 8;
          const float4 d0 = *reinterpret_cast<const float4*>(scr + erl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + erl * CP + c0 + 4);
          const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
          const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          const uint32_t uw[4] = {eur[pass][j].x, eur[pass][j].y, eur[pass][j].z, eur[pass][j].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
            const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
            s1 += t;
            s2 = fmaf(t, xh, s2);
          }
        }
        s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
        s1 *= (1.0f / C); s2 *= (1.0f / C);
#pragma unroll
        for (int j = 0; j < ENJ; ++j) {
          const int c0 = (eq + 4 * j) * 8;
          const float4 d0 = *reinterpret_cast<const float4*>(scr + erl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + erl * CP + c0 + 4);
          const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
          const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          const uint32_t uw[4] = {eur[pass][j].x, eur[pass][j].y, eur[pass][j].z, eur[pass][j].w};
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
            const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
            o[e] = rstd * (t - s1 - xh * s2);
          }
          if (m < p.M)
            *reinterpret_cast<uint4*>(p.da + m * C + c0) = make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
        }
        if (pass == 0) __syncthreads();                               // the scratch rows are free for the second pass
      }
    } else {
      const float* scr = reinterpret_cast<const float*>(lds) + pair * (16 * C);
      constexpr int NCH = 16 * C / 8 / 64;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        const long e0 = (m0 + 16 * pass) * C;
        const long e_end = p.M * C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int idx = k * 64 + lane;
          const long e = e0 + idx * 8;
          const float4 d0 = reinterpret_cast<const float4*>(scr)[2 * idx], d1 = reinterpret_cast<const float4*>(scr)[2 * idx + 1];
          if (e < e_end)
            *reinterpret_cast<uint4*>(p.da + e) = make_uint4(pack_bf16(d0.x, d0.y), pack_bf16(d0.z, d0.w), pack_bf16(d1.x, d1.y), pack_bf16(d1.z, d1.w));
        }
        if (pass == 0) __syncthreads();
      }
    }
#endif
    return;
  }

  // ================================================================== consumer: weight DMA, GEMM3, epilogue
  if constexpr (BLK2_PRIO == 2) __builtin_amdgcn_s_setprio(2);
  f32x16 acc3[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[cb][r] = 0.f;
#pragma unroll
  for (int i = 0; i < R1W; ++i) DMA_A_PIECE(0, i)
#pragma unroll
  for (int i = 0; i < R1W; ++i) DMA_A_PIECE(1, i)
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  constexpr int PFC = 4, DMA_EVERY = 2 * G::CB / NDMA;
  static_assert(DMA_EVERY >= 1 && DMA_EVERY * NDMA <= 2 * G::CB, "one DMA instruction per DMA_EVERY MFMAs");
  // fragment j of a block's GEMM3 stream -> piece of the ring slot: (t, cb) order (consecutive MFMAs update different accumulators) out of
  // the packed (cb, t) order
#define G3_PIECE(J) ((((J) % G::CB) * 2) + ((J) / G::CB))
#define CB_SYNC(B, ST)                                                                                     \
  if (ST || (B) + 2 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R1W) : "memory");                   \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
  __builtin_amdgcn_s_barrier();
#define CB_BLOCK(B, ST)                                                                                    \
  {                                                                                                        \
    if (ST || (B) < G::NHB) {                              /* the Hpre tile of block B for the producer's block B + 1 */ \
      glds16(hp_base + static_cast<long>(B) * 2048, lane16, hp_lds + ((B) & 1) * 4096);                    \
      glds16(hp_base + static_cast<long>(B) * 2048 + 1024, lane16, hp_lds + ((B) & 1) * 4096 + 1024);      \
    }                                                                                                      \
    if (ST || ((B) >= 2 && (B) - 2 < G::NHB)) {                                                            \
      const uint4* hr = reinterpret_cast<const uint4*>(hb_lane + (((B) - 2) & 1) * 4096);                  \
      const uint4 hq0 = hr[0], hq1 = hr[1];                                                                \
      const bf16x8 hf0 = __builtin_bit_cast(bf16x8, hq0), hf1 = __builtin_bit_cast(bf16x8, hq1);           \
      if constexpr (EMIT == 2) {   /* dHpre of block B - 2 in its Hpre's tile (CNX_TN_ACC), in front of this block's DMA */ \
        uint4* dd_ = reinterpret_cast<uint4*>(p.dhpt_out) + (tile * G::NHB + ((B) - 2)) * 128 + l32 * 4 + half * 2; \
        dd_[0] = hq0; dd_[1] = hq1;                                                                        \
      }                                                                                                    \
      const unsigned char* sl = lds + G::W1_RING + (((B) - 2) % 2) * (2 * G::CB * 1024) + lane * 16;       \
      bf16x8 fr[PFC];                                                                                      \
      _Pragma("unroll") for (int j = 0; j < PFC; ++j) fr[j] = *reinterpret_cast<const bf16x8*>(sl + G3_PIECE(j) * 1024); \
      _Pragma("unroll") for (int j = 0; j < 2 * G::CB; ++j) {                                              \
        acc3[j % G::CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(j < G::CB ? hf0 : hf1, fr[j % PFC], acc3[j % G::CB], 0, 0, 0); \
        if (j + PFC < 2 * G::CB) fr[j % PFC] = *reinterpret_cast<const bf16x8*>(sl + G3_PIECE(j + PFC) * 1024); \
        if (j % DMA_EVERY == 0 && j / DMA_EVERY < NDMA) { BLK_DMA(B, j / DMA_EVERY, ST) }                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
      }                                                                                                    \
    } else {                                                                                               \
      _Pragma("unroll") for (int k = 0; k < NDMA; ++k) { BLK_DMA(B, k, false) }                            \
    }                                                                                                      \
    CB_SYNC(B, ST)                                                                                         \
  }
  CB_BLOCK(0, false)
  CB_BLOCK(1, false)
  for (int b = 2; b + 2 < G::NHB; ++b) CB_BLOCK(b, true)
  CB_BLOCK(G::NHB - 2, false)
  CB_BLOCK(G::NHB - 1, false)
  CB_BLOCK(G::NHB, false)
  CB_BLOCK(G::NHB + 1, false)
#undef CB_BLOCK
#undef CB_SYNC
#undef G3_PIECE
#undef BLK_DMA
#undef DMA_A_PIECE
#undef DMA_B_PIECE
  // ---- epilogue: acc3[cb][r] = da[m0 + (r&3) + 8*(r>>2) + 4*half][cb*32 + l32], through the dead rings, 16 rows per pass (as
  //      blk_mlp_bwd_kernel)
  __syncthreads();
#if BLK2_EPI
  {
    // this wavefront only scatters; the producer of the pair - which holds the u rows - does the LayerNorm backward and the stores
    constexpr int CPS = LNB ? C + 4 : C;                                 // (LNB: padded rows, the 4 lanes x 16 rows spread over the banks)
    static_assert(4 * 16 * CPS * 4 <= G::W1_RING + G::W2_RING, "the epilogue tile reuses the weight rings");
    float* scr = reinterpret_cast<float*>(lds) + pair * (16 * CPS);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          scr[((r & 3) + 8 * (r >> 2) + 4 * half) * CPS + cb * 32 + l32] = acc3[cb][8 * pass + r];
      __syncthreads();                                                // scattered: the producer reads
      if (pass == 0) __syncthreads();                                 // ... and is done with these rows
    }
  }
#else
  if constexpr (LNB) {
    constexpr int CP = C + 4;
    static_assert(4 * 16 * CP * 4 <= G::W1_RING + G::W2_RING, "the epilogue tile reuses the weight rings");
    float* scr = reinterpret_cast<float*>(lds) + pair * (16 * CP);
    constexpr int NJ = C / 32;
    const int rl = lane >> 2, q = lane & 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          scr[((r & 3) + 8 * (r >> 2) + 4 * half) * CP + cb * 32 + l32] = acc3[cb][8 * pass + r];
      __builtin_amdgcn_wave_barrier();
      const long m = m0 + 16 * pass + rl;
      const long mc = m < p.M ? m : p.M - 1;
      const float mean = p.mean[mc], rstd = p.rstd[mc];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = (q + 4 * j) * 8;
        const uint4 ur = *reinterpret_cast<const uint4*>(p.u + mc * C + c0);
        const float4 d0 = *reinterpret_cast<const float4*>(scr + rl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + rl * CP + c0 + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const uint32_t uw[4] = {ur.x, ur.y, ur.z, ur.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
          const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
          s1 += t;
          s2 = fmaf(t, xh, s2);
        }
      }
      s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
      s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
      s1 *= (1.0f / C); s2 *= (1.0f / C);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c0 = (q + 4 * j) * 8;
        const uint4 ur = *reinterpret_cast<const uint4*>(p.u + mc * C + c0);
        const float4 d0 = *reinterpret_cast<const float4*>(scr + rl * CP + c0), d1 = *reinterpret_cast<const float4*>(scr + rl * CP + c0 + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + c0), w1 = *reinterpret_cast<const float4*>(p.ln_w + c0 + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const uint32_t uw[4] = {ur.x, ur.y, ur.z, ur.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float uv = (e & 1) ? bf16_hi(uw[e >> 1]) : bf16_lo(uw[e >> 1]);
          const float t = wv[e] * dv[e], xh = (uv - mean) * rstd;
          o[e] = rstd * (t - s1 - xh * s2);
        }
        if (m < p.M)
          *reinterpret_cast<uint4*>(p.da + m * C + c0) = make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
      }
    }
  } else {
    float* scr = reinterpret_cast<float*>(lds) + pair * (16 * C);
    constexpr int NCH = 16 * C / 8 / 64;                                  // 8-element chunks per lane and pass (C / 32)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const long e0 = (m0 + 16 * pass) * C;
      const long e_end = p.M * C;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          scr[((r & 3) + 8 * (r >> 2) + 4 * half) * C + cb * 32 + l32] = acc3[cb][8 * pass + r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int idx = k * 64 + lane;
        const long e = e0 + idx * 8;
        const float4 d0 = reinterpret_cast<const float4*>(scr)[2 * idx], d1 = reinterpret_cast<const float4*>(scr)[2 * idx + 1];
        if (e < e_end)
          *reinterpret_cast<uint4*>(p.da + e) = make_uint4(pack_bf16(d0.x, d0.y), pack_bf16(d0.z, d0.w), pack_bf16(d1.x, d1.y), pack_bf16(d1.z, d1.w));
      }
    }
  }
#endif
}

// Which Hpre backward serves width C: the wavefront-pair kernel at C = 256 / 384 (cnx_runtime_switch(CNX_SWITCH_BLK2_BWD_WIDTHS) /
// APGD_BLK2B: bit 0 = C 256, bit 1 = C 384)
int& blk2b_widths() {
  static int m = [] {
    const char* env = getenv("APGD_BLK2B");
    return env ? ((strstr(env, "256") ? 1 : 0) | (strstr(env, "384") ? 2 : 0)) : 3;
  }();
  return m;
}

template <int C>
int launch_blk2_bwd_hpre(const BlkBwdArgs& a, int g_dtype, hipStream_t s) {
  using G = Geo2<C>;
  const dim3 grid(static_cast<unsigned>((a.M + 127) / 128)), block(512);
#define BLK2B_GO(TG, EM, LN)                                                                                     \
  {                                                                                                              \
    auto kfn = blk2_bwd_kernel<C, TG, EM, LN>;                                                                   \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS); \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, G::LDS, s, a);                                                          \
  }
#define BLK2B_LAUNCH(TG) { if (a.dhpt_out && a.u) BLK2B_GO(TG, 2, true) else if (a.dhpt_out) BLK2B_GO(TG, 2, false) else BLK2B_GO(TG, 0, true) }
  if (g_dtype == APGD_F32) BLK2B_LAUNCH(float) else BLK2B_LAUNCH(uint16_t)
#undef BLK2B_LAUNCH
#undef BLK2B_GO
  return launch_status();
}

template <int C>
int launch_blk_bwd_hpre(const BlkBwdArgs& a, int g_dtype, hipStream_t s) {
  using G = GeoB<C>;
  if constexpr (C == 256 || C == 384) {
    if (blk2b_widths() & (C == 256 ? 1 : 2)) return launch_blk2_bwd_hpre<C>(a, g_dtype, s);
  }
  constexpr int LDS_BYTES = G::DEPTH * (G::KS + 2 * G::CB) * 1024 + 16 * C;
  const dim3 grid(static_cast<unsigned>((a.M + G::BM - 1) / G::BM)), block(G::WAVES * 64);
#define BLK_LAUNCH(TG)                                                                                           \
  if (a.dhpt_out && a.u) {                                               /* training backward, LayerNorm backward in the epilogue */ \
    auto kfn = blk_mlp_bwd_kernel<C, TG, 2, true, true>;                                                         \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, LDS_BYTES, s, a);                                                       \
  } else if (a.dhpt_out) {                                                                                       \
    auto kfn = blk_mlp_bwd_kernel<C, TG, 2, false, true>;                                                        \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, LDS_BYTES, s, a);                                                       \
  } else {                                                                                                       \
    auto kfn = blk_mlp_bwd_kernel<C, TG, 0, true, true>;                                                         \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, LDS_BYTES, s, a);                                                       \
  }
  if (g_dtype == APGD_F32) BLK_LAUNCH(float) else BLK_LAUNCH(uint16_t)
#undef BLK_LAUNCH
  return launch_status();
}

}  // namespace

extern "C" {

int cnx_runtime_switch(int32_t which, int32_t value) {
  switch (which) {
    case CNX_SWITCH_BLK2_WIDTHS: {
      int& m = blk2_widths();
      const int prev = m;
      if (value >= 0) m = value & 3;
      return prev;
    }
    case CNX_SWITCH_DW_SHARED_HALO: return dw_shared_halo_switch(value);
    case CNX_SWITCH_BLK2_BWD_WIDTHS: {
      int& m = blk2b_widths();
      const int prev = m;
      if (value >= 0) m = value & 3;
      return prev;
    }
    case CNX_SWITCH_FWD_WAVES8: {
      if (!BLK_FWD_W8_BUILD) return -1;
      int& m = fwd_w8_widths();
      const int prev = m;
      if (value >= 0) m = value & 3;
      return prev;
    }
    default: return -1;
  }
}

int cnx_block_mlp_supported(int32_t C) { return (C == 96 || C == 128 || C == 192 || C == 256 || C == 384) ? 1 : 0; }

int64_t cnx_mlp_packed_elems(int32_t C) { return static_cast<int64_t>(8) * C * C + (blk_fwd_pipe(C) ? 64 * C : 0); }

int cnx_mlp_pack_weights(const void* W1, const void* W2, int w_dtype, void* Wf, int32_t C, void* stream) {
  if (C <= 0 || C % 32 != 0) return APGD_ERR_SIZE;
  if (!W1 || !W2 || !Wf) return APGD_ERR_NULL;
  const long total = static_cast<long>(C / 8 + (blk_fwd_pipe(C) ? 1 : 0)) * (C / 16 + 2 * (C / 32)) * 64;
  const dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
  hipStream_t s = as_stream(stream);
  if (w_dtype == APGD_F32)
    hipLaunchKernelGGL(pack_fwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(W1), static_cast<const float*>(W2),
                       static_cast<uint16_t*>(Wf), C);
  else if (w_dtype == APGD_BF16)
    hipLaunchKernelGGL(pack_fwd_kernel<__bf16>, grid, block, 0, s, static_cast<const __bf16*>(W1),
                       static_cast<const __bf16*>(W2), static_cast<uint16_t*>(Wf), C);
  else return APGD_ERR_DTYPE;
  return launch_status();
}

static int block_mlp_fwd_impl(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                              const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                              int resid_dtype, void* out, int out_dtype, void* y2_out, void* hpre_ws, int64_t M, int32_t C,
                              void* stream, void* h_ws = nullptr, void* a_rows = nullptr) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!u || !Wf || !b1 || !b2 || !out) return APGD_ERR_NULL;
  if (ln_w && !ln_b) return APGD_ERR_NULL;
  if ((mean == nullptr) != (rstd == nullptr)) return APGD_ERR_NULL;
  if ((resid_dtype != APGD_F32 && resid_dtype != APGD_BF16) || (out_dtype != APGD_F32 && out_dtype != APGD_BF16))
    return APGD_ERR_DTYPE;
  BlkFwdArgs a;
  a.u = static_cast<const uint16_t*>(u); a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps; a.mean = mean; a.rstd = rstd;
  a.Wf = static_cast<const uint16_t*>(Wf); a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.resid = resid; a.out = out;
  a.y2 = static_cast<uint16_t*>(y2_out); a.M = M;
  a.hpre = static_cast<uint16_t*>(hpre_ws);
  a.hact = static_cast<uint16_t*>(h_ws); a.a_out = static_cast<uint16_t*>(a_rows);
  if ((h_ws != nullptr) != (a_rows != nullptr) || (h_ws && !hpre_ws)) return APGD_ERR_NULL;
  if (hpre_ws && !blk_fwd_pipe(C)) return APGD_ERR_ARG;          // only the pipelined loop writes the workspace
#if MLP_ABLATE
  static const int dbg = getenv("APGD_BLK_DBG") ? atoi(getenv("APGD_BLK_DBG")) : 0;   // timing experiments: ablation builds only
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  hipStream_t s = as_stream(stream);
  switch (C) {
    case 96: return launch_blk_fwd<96>(a, resid_dtype, out_dtype, s);
    case 128: return launch_blk_fwd<128>(a, resid_dtype, out_dtype, s);
    case 192: return launch_blk_fwd<192>(a, resid_dtype, out_dtype, s);
    case 256: return use_blk2(256, a.hpre != nullptr) ? launch_blk2_fwd<256>(a, resid_dtype, out_dtype, s) : launch_blk_fwd<256>(a, resid_dtype, out_dtype, s);
    case 384: return use_blk2(384, a.hpre != nullptr) ? launch_blk2_fwd<384>(a, resid_dtype, out_dtype, s) : launch_blk_fwd<384>(a, resid_dtype, out_dtype, s);
    default: return APGD_ERR_ARG;
  }
}

int cnx_block_mlp_fwd(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                      const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                      int resid_dtype, void* out, int out_dtype, void* y2_out, int64_t M, int32_t C, void* stream) {
  return block_mlp_fwd_impl(u, ln_w, ln_b, eps, mean, rstd, Wf, b1, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, nullptr,
                            M, C, stream);
}

int cnx_block_mlp_hpre_supported(int32_t C) { return (C == 128 || C == 192 || C == 256 || C == 384) ? 1 : 0; }

int64_t cnx_block_mlp_hpre_elems(int64_t M, int32_t C) { return M <= 0 ? 0 : ((M + 127) / 128) * 128 * 4 * static_cast<int64_t>(C); }

int cnx_block_mlp_fwd_hpre(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                           const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                           int resid_dtype, void* out, int out_dtype, void* hpre_ws, int64_t M, int32_t C, void* stream) {
  if (!hpre_ws) return APGD_ERR_NULL;
  if (!cnx_block_mlp_hpre_supported(C)) return APGD_ERR_ARG;
  return block_mlp_fwd_impl(u, ln_w, ln_b, eps, mean, rstd, Wf, b1, b2, gamma, resid, resid_dtype, out, out_dtype, nullptr, hpre_ws,
                            M, C, stream);
}

int cnx_block_mlp_bwd_input_hpre(const void* u, const float* ln_w, const float* mean, const float* rstd, const void* g,
                                 int g_dtype, const float* gamma, const void* Wb, const void* hpre_ws, void* du, int64_t M,
                                 int32_t C, void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!u || !ln_w || !mean || !rstd || !g || !Wb || !hpre_ws || !du) return APGD_ERR_NULL;
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  BlkBwdArgs a;
  a.u = static_cast<const uint16_t*>(u); a.ln_w = ln_w; a.ln_b = nullptr; a.mean = mean; a.rstd = rstd; a.g = g; a.gamma = gamma;
  a.Wb = static_cast<const uint16_t*>(Wb); a.b1 = nullptr; a.da = static_cast<uint16_t*>(du);
  a.a_out = a.do_out = a.ht_out = a.dhpt_out = nullptr; a.hpre = static_cast<const uint16_t*>(hpre_ws); a.M = M; a.a_stride = C; a.emit_acc = 0;
  switch (C) {
    case 128: return launch_blk_bwd_hpre<128>(a, g_dtype, as_stream(stream));
    case 192: return launch_blk_bwd_hpre<192>(a, g_dtype, as_stream(stream));
    case 256: return launch_blk_bwd_hpre<256>(a, g_dtype, as_stream(stream));
    case 384: return launch_blk_bwd_hpre<384>(a, g_dtype, as_stream(stream));
    default: return APGD_ERR_ARG;
  }
}

int cnx_block_mlp_fwd_train(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                            const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                            int resid_dtype, void* out, int out_dtype, void* y2_out, void* hpre_ws, void* h_ws, void* a_rows,
                            int64_t M, int32_t C, void* stream) {
  if (!hpre_ws || !h_ws || !a_rows || !ln_w) return APGD_ERR_NULL;
  if (!cnx_block_mlp_hpre_supported(C)) return APGD_ERR_ARG;
  return block_mlp_fwd_impl(u, ln_w, ln_b, eps, mean, rstd, Wf, b1, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, hpre_ws,
                            M, C, stream, h_ws, a_rows);
}

int cnx_block_mlp_bwd_train_hpre(const void* g, int g_dtype, const float* gamma, const void* Wb, const void* hpre_ws, void* da,
                                 void* do_rows, void* dhpre_ws, int64_t M, int32_t C, void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!g || !Wb || !hpre_ws || !da || !do_rows || !dhpre_ws) return APGD_ERR_NULL;
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  BlkBwdArgs a;
  a.u = nullptr; a.ln_w = nullptr; a.ln_b = nullptr; a.mean = nullptr; a.rstd = nullptr; a.g = g; a.gamma = gamma;
  a.Wb = static_cast<const uint16_t*>(Wb); a.b1 = nullptr; a.da = static_cast<uint16_t*>(da);
  a.a_out = nullptr; a.do_out = static_cast<uint16_t*>(do_rows); a.ht_out = nullptr; a.dhpt_out = static_cast<uint16_t*>(dhpre_ws);
  a.hpre = static_cast<const uint16_t*>(hpre_ws); a.M = M; a.a_stride = C; a.emit_acc = 1;
  switch (C) {
    case 128: return launch_blk_bwd_hpre<128>(a, g_dtype, as_stream(stream));
    case 192: return launch_blk_bwd_hpre<192>(a, g_dtype, as_stream(stream));
    case 256: return launch_blk_bwd_hpre<256>(a, g_dtype, as_stream(stream));
    case 384: return launch_blk_bwd_hpre<384>(a, g_dtype, as_stream(stream));
    default: return APGD_ERR_ARG;
  }
}

int cnx_block_mlp_bwd_train_hpre_ln(const void* u, const float* ln_w, const float* mean, const float* rstd, const void* g, int g_dtype,
                                    const float* gamma, const void* Wb, const void* hpre_ws, void* du, void* do_rows, void* dhpre_ws,
                                    int64_t M, int32_t C, void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!u || !ln_w || !mean || !rstd || !g || !Wb || !hpre_ws || !du || !do_rows || !dhpre_ws) return APGD_ERR_NULL;
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  BlkBwdArgs a;
  a.u = static_cast<const uint16_t*>(u); a.ln_w = ln_w; a.ln_b = nullptr; a.mean = mean; a.rstd = rstd; a.g = g; a.gamma = gamma;
  a.Wb = static_cast<const uint16_t*>(Wb); a.b1 = nullptr; a.da = static_cast<uint16_t*>(du);
  a.a_out = nullptr; a.do_out = static_cast<uint16_t*>(do_rows); a.ht_out = nullptr; a.dhpt_out = static_cast<uint16_t*>(dhpre_ws);
  a.hpre = static_cast<const uint16_t*>(hpre_ws); a.M = M; a.a_stride = C; a.emit_acc = 1;
  switch (C) {
    case 128: return launch_blk_bwd_hpre<128>(a, g_dtype, as_stream(stream));
    case 192: return launch_blk_bwd_hpre<192>(a, g_dtype, as_stream(stream));
    case 256: return launch_blk_bwd_hpre<256>(a, g_dtype, as_stream(stream));
    case 384: return launch_blk_bwd_hpre<384>(a, g_dtype, as_stream(stream));
    default: return APGD_ERR_ARG;
  }
}

int64_t cnx_mlp_packed_bwd_elems(int32_t C) { return static_cast<int64_t>(12) * C * C; }

int cnx_mlp_pack_weights_bwd(const void* W1, const void* W2, int w_dtype, void* Wb, int32_t C, void* stream) {
  if (C <= 0 || C % 32 != 0) return APGD_ERR_SIZE;
  if (!W1 || !W2 || !Wb) return APGD_ERR_NULL;
  const long total = static_cast<long>(C / 8) * (2 * (C / 16) + 2 * (C / 32)) * 64;
  const dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
  hipStream_t s = as_stream(stream);
  if (w_dtype == APGD_F32)
    hipLaunchKernelGGL(pack_bwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(W1), static_cast<const float*>(W2),
                       static_cast<uint16_t*>(Wb), C);
  else if (w_dtype == APGD_BF16)
    hipLaunchKernelGGL(pack_bwd_kernel<__bf16>, grid, block, 0, s, static_cast<const __bf16*>(W1),
                       static_cast<const __bf16*>(W2), static_cast<uint16_t*>(Wb), C);
  else return APGD_ERR_DTYPE;
  return launch_status();
}

static int block_mlp_bwd_impl(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                              const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* da,
                              void* a_out, int64_t a_stride, void* do_out, void* ht_out, void* dhpt_out, bool ln_bwd,
                              int64_t M, int32_t C, void* stream, int emit_acc = 0) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (emit_acc && M % 32 != 0) return APGD_ERR_ARG;                  // whole tiles
  if (M == 0) return APGD_OK;
  if (!u || !ln_w || !ln_b || !mean || !rstd || !g || !Wb || !b1 || !da) return APGD_ERR_NULL;
  const int n_emit = (a_out != nullptr) + (do_out != nullptr) + (ht_out != nullptr) + (dhpt_out != nullptr);
  if (n_emit != 0 && n_emit != 4) return APGD_ERR_NULL;
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  BlkBwdArgs a;
  a.u = static_cast<const uint16_t*>(u); a.ln_w = ln_w; a.ln_b = ln_b; a.mean = mean; a.rstd = rstd; a.g = g; a.gamma = gamma;
  a.Wb = static_cast<const uint16_t*>(Wb); a.b1 = b1; a.da = static_cast<uint16_t*>(da);
  a.a_out = static_cast<uint16_t*>(a_out); a.do_out = static_cast<uint16_t*>(do_out);
  a.ht_out = static_cast<uint16_t*>(ht_out); a.dhpt_out = static_cast<uint16_t*>(dhpt_out); a.hpre = nullptr; a.M = M; a.emit_acc = emit_acc;
  if (a_stride != 0 && (a_stride < C || a_stride % 8 != 0)) return APGD_ERR_ARG;
  a.a_stride = a_stride ? a_stride : C;
  hipStream_t s = as_stream(stream);
  switch (C) {
    case 96: return launch_blk_bwd<96>(a, g_dtype, ln_bwd, s);
    case 128: return launch_blk_bwd<128>(a, g_dtype, ln_bwd, s);
    case 192: return launch_blk_bwd<192>(a, g_dtype, ln_bwd, s);
    case 256: return launch_blk_bwd<256>(a, g_dtype, ln_bwd, s);
    default: return APGD_ERR_ARG;
  }
}

int cnx_block_mlp_bwd(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                      const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* da,
                      void* a_out, int64_t a_stride, void* do_out, void* ht_out, void* dhpt_out, int64_t M, int32_t C,
                      void* stream) {
  return block_mlp_bwd_impl(u, ln_w, ln_b, mean, rstd, g, g_dtype, gamma, Wb, b1, da, a_out, a_stride, do_out, ht_out, dhpt_out,
                            false, M, C, stream);
}

int cnx_block_mlp_bwd_input(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                            const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* du,
                            int64_t M, int32_t C, void* stream) {
  return block_mlp_bwd_impl(u, ln_w, ln_b, mean, rstd, g, g_dtype, gamma, Wb, b1, du, nullptr, 0, nullptr, nullptr, nullptr,
                            true, M, C, stream);
}

int cnx_block_mlp_bwd_acc(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                          const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* da,
                          void* a_rows, void* do_rows, void* h_ws, void* dhpre_ws, int64_t M, int32_t C, void* stream) {
  if (!a_rows || !do_rows || !h_ws || !dhpre_ws) return APGD_ERR_NULL;
  return block_mlp_bwd_impl(u, ln_w, ln_b, mean, rstd, g, g_dtype, gamma, Wb, b1, da, a_rows, 0, do_rows, h_ws, dhpre_ws, false, M, C,
                            stream, 1);
}

int cnx_block_mlp_bwd_acc_ln(const void* u, const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                             const void* g, int g_dtype, const float* gamma, const void* Wb, const float* b1, void* du,
                             void* a_rows, void* do_rows, void* h_ws, void* dhpre_ws, int64_t M, int32_t C, void* stream) {
  if (!a_rows || !do_rows || !h_ws || !dhpre_ws) return APGD_ERR_NULL;
  return block_mlp_bwd_impl(u, ln_w, ln_b, mean, rstd, g, g_dtype, gamma, Wb, b1, du, a_rows, 0, do_rows, h_ws, dhpre_ws, true, M, C,
                            stream, 1);
}

int cnx_block_mlp_bwd_supported(int32_t C) { return (C == 96 || C == 128 || C == 192 || C == 256) ? 1 : 0; }

}  // extern "C"

#if MLP_ABLATE
extern "C" int cnx_dbg_blk_trace(unsigned long long* host, int n_wgs) {
  if (n_wgs > BLK_TRACE_WGS) n_wgs = BLK_TRACE_WGS;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_blk_trace), sizeof(unsigned long long) * BLK_TRACE_SLOTS * n_wgs));
}
#endif
