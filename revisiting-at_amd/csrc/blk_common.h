// blk_common.h - what the two translation units of the fused LN + MLP kernels share (block_kernels.hip: forward kernels; block_bwd_kernels.hip:
// backward kernels): build switches, vector types, the LDS-DMA helper, the activation arithmetic (GELU / GELU' in packed and unpacked form) and
// the geometry of the wavefront-pair kernels.  Everything lives in an anonymous namespace: internal to each translation unit.  Not part of the C ABI.
#pragma once
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
// measurement builds only (`make EXTRA=-DBLK2_C192_BUILD=1`): the wavefront-pair kernels at C = 192 - bit-identical and 10 - 32 % SLOWER than the
// two independent four-wavefront workgroups per CU that serve that width (profiles/r06_fused_mlp.md section 1b); not instantiated otherwise
// operand fragments a pair's producer / consumer keeps in flight ahead of its MFMAs (LDS reads issued PF fragments early)
#ifndef BLK2_PF
#define BLK2_PF 4
#endif
#ifndef BLK2_PFC
#define BLK2_PFC 4
#endif
#ifndef BLK2_C192_BUILD
#define BLK2_C192_BUILD 0
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


template <int C>
struct Geo2 {
  static constexpr int KS = C / 16, CB = C / 32, NHB = C / 8;
  static constexpr int PIECES = KS + 2 * CB, SLICE = PIECES * 1024;
  // a block's KS + 2 CB weight pieces are dealt over the eight wavefronts, piece K * 8 + wavefront for the wavefront's K-th instruction: first
  // the 2 CB pieces of the second ring, then the KS of the first.  (C a multiple of 64: the type of a wavefront's K-th piece is then the same
  // for the four wavefronts of a role, i.e. a compile-time property of the role's code; at C = 192 the producers move 2 + 1, the consumers 1 + 2.)
  static constexpr int NDMA = (KS + 2 * CB) / 8;
  static_assert((KS + 2 * CB) % 8 == 0 && (2 * CB) % 4 == 0, "pieces deal evenly over eight wavefronts");
  static constexpr int r1w(int role) {                                // first-ring pieces among a wavefront's NDMA instructions (the LAST ones)
    int n = 0;
    for (int k = 0; k < NDMA; ++k) n += (k * 8 + role * 4 >= 2 * CB) ? 1 : 0;
    return n;
  }
  static constexpr int W1_RING = 3 * KS * 1024, W2_RING = 2 * 2 * CB * 1024, HBUF = 4 * 2 * 4096;   // per pair, two buffers of [H | Hpre] tiles
  static constexpr int LDS = W1_RING + W2_RING + HBUF + 16 * C;
  static_assert(LDS <= 160 * 1024 && 2 * C * 4 <= HBUF && 4 * 16 * C * 4 <= W1_RING + W2_RING, "LDS plan");
};

}  // namespace
