// block_bwd_kernels.hip - backward kernels of the fused ConvNeXt block tail for gfx950 (MI355X): the input gradient of
//     out = x + gamma * ( GELU( LN(u) W1^T + b1 ) W2^T + b2 )
// (/root/reference/models/convnext.py:40-49 backward) in its recomputing, Hpre-workspace and emitting (training) forms, on one wavefront per
// row tile (blk_mlp_bwd_kernel) and on wavefront pairs (blk2_bwd_kernel, round 6).  Split from block_kernels.hip in round 6 (one translation
// unit had grown to six minutes of compile time); the forward kernels, the work decomposition and the K-index conventions are described there.
#include "blk_common.h"

namespace {

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
#define DMA_A_PIECE(T, Q)                                                                                  \
  {                                                                                                        \
    const int q_ = (Q);                                                                                    \
    glds16(wsrc + static_cast<long>(T) * SLICE_B + (G::KS + q_) * 1024, lane16, ring1 + ((T) % 3) * (G::KS * 1024) + q_ * 1024); \
  }
#define DMA_B_PIECE(T, Q)                                                                                  \
  {                                                                                                        \
    const int q_ = (Q);                                                                                    \
    glds16(wsrc + static_cast<long>(T) * SLICE_B + (2 * G::KS + q_) * 1024, lane16, ring2 + ((T) % 2) * (2 * G::CB * 1024) + q_ * 1024); \
  }
  constexpr int NDMA = G::NDMA, R1W_ALL = (G::KS + 7) / 8;
  const int w4 = wave & 3;
  // block B, DMA instruction K of this wavefront (piece K * 8 + wavefront of the block's list, as blk2_fwd_kernel): first its GEMM3 pieces of
  // block B - 1 (read in block B + 1), then its W2^T pieces of block B + 2 (read in block B + 2)
#define BLK_DMA(B, K, ST)                                                                                  \
  if ((K) * 8 + ROLE * 4 < 2 * G::CB) { if (ST || ((B) >= 1 && (B) - 1 < G::NHB)) DMA_B_PIECE((B) - 1, (K) * 8 + ROLE * 4 + w4) } \
  else { if (ST || (B) + 2 < G::NHB) DMA_A_PIECE((B) + 2, (K) * 8 + ROLE * 4 + w4 - 2 * G::CB) }
#define DMA_A_ALL(T)                                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < R1W_ALL; ++i_) {                                                 \
    if (G::KS % 8 == 0 || i_ * 8 + wave < G::KS) DMA_A_PIECE(T, i_ * 8 + wave)                             \
  }
  const long tile = static_cast<long>(blockIdx.x) * 4 + pair;
  unsigned char* hb_lane = hbuf + pair * 8192 + lane * 32;           // buffer i at + 4096 i: the dHpre tile of a block
  // the Hpre tile of block b ((tile NHB + b) 2048 bytes into the workspace) goes, as it lies, into the second half of buffer b & 1: issued
  // by the CONSUMER at the top of its block b (it has the lighter instruction stream), read by the producer at the top of block b + 1
  const unsigned char* hp_base = reinterpret_cast<const unsigned char*>(p.hpre) + tile * G::NHB * 2048;
  const uint32_t hp_lds = __builtin_amdgcn_readfirstlane(lds_addr(hbuf)) + pair * 8192 + 2048;

  if (wave < 4) {
    // ================================================================ producer: dO rows, dH, GELU', dHpre -> LDS
    constexpr int ROLE = 0, R1W = G::r1w(ROLE);
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
    DMA_A_ALL(0)
    DMA_A_ALL(1)
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // W2^T(0), W2^T(1): this wavefront's pieces (and its dO row stores)
    __syncthreads();                                                  // ... everybody's
    float c6v = 1.8761737253e-03f;
    asm volatile("" : "+v"(c6v));
    // the lane's 16 values of a Hpre tile: + 64 l32 + 32 half of the copy in the hand-over buffer
    const unsigned char* hp_lane = hbuf + pair * 8192 + 2048 + l32 * 64 + half * 32;
    // Block b:  MFMA stream  dH(b) = W2[:, b]^T x dO^T (KS MFMAs, b < NHB); the Hpre(b) tile is requested at the top
    //           VALU stream  dHpre(b - 1) = dH(b - 1) * GELU'(Hpre(b - 1)) between those MFMAs (b >= 1) -> hand-over buffer
    constexpr int PF = BLK2_PF, NUOP = 4 * 62;
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
  constexpr int ROLE = 1, R1W = G::r1w(ROLE);
  if constexpr (BLK2_PRIO == 2) __builtin_amdgcn_s_setprio(2);
  f32x16 acc3[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[cb][r] = 0.f;
  DMA_A_ALL(0)
  DMA_A_ALL(1)
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  constexpr int PFC = BLK2_PFC, DMA_EVERY = 2 * G::CB / NDMA;
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
#undef DMA_A_ALL
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
    return env ? ((strstr(env, "256") ? 1 : 0) | (strstr(env, "384") ? 2 : 0) | (strstr(env, "192") ? 4 : 0)) : 3;
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
  if constexpr ((BLK2_C192_BUILD && C == 192) || C == 256 || C == 384) {
    if (blk2b_widths() & (C == 256 ? 1 : C == 384 ? 2 : 4)) return launch_blk2_bwd_hpre<C>(a, g_dtype, s);
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

// (the switch behind cnx_runtime_switch(CNX_SWITCH_BLK2_BWD_WIDTHS, .), whose entry point lives with the forward kernels)
int blk2b_widths_switch(int value) {
  int& m = blk2b_widths();
  const int prev = m;
  if (value >= 0) m = value & 7;
  return prev;
}

extern "C" {

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
