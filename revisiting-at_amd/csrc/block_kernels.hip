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
#include "blk_common.h"

namespace {

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
#define DMA_W1_PIECE(T, Q)                                                                                 \
  {                                                                                                        \
    const int q_ = (Q);                                                                                    \
    glds16(wsrc + static_cast<long>(T) * G::SLICE + q_ * 1024, lane16, ring1 + ((T) % 3) * (G::KS * 1024) + q_ * 1024); \
  }
#define DMA_W2_PIECE(T, Q)                                                                                 \
  {                                                                                                        \
    const int q_ = (Q);                                                                                    \
    glds16(wsrc + static_cast<long>((T) + 1) * G::SLICE + (G::KS + q_) * 1024, lane16, ring2 + ((T) % 2) * (2 * G::CB * 1024) + q_ * 1024); \
  }
  // block B, DMA instruction K of this wavefront (piece K * 8 + wavefront of the block's list): first its W2(B - 1) pieces (read in block
  // B + 1; the slot W2(B - 3) left), then its W1(B + 2) pieces (read in block B + 2; the slot W1(B - 1) left).  ROLE (0 producer, 1 consumer)
  // and w4 = wavefront % 4 are defined by the role's code: whether instruction K is a W2 or a W1 piece is a constant there
  constexpr int NDMA = G::NDMA, R1W_ALL = (G::KS + 7) / 8;
  const int w4 = wave & 3;
#define BLK_DMA(B, K, ST)                                                                                  \
  if ((K) * 8 + ROLE * 4 < 2 * G::CB) { if (ST || ((B) >= 1 && (B) - 1 < G::NHB)) DMA_W2_PIECE((B) - 1, (K) * 8 + ROLE * 4 + w4) } \
  else { if (ST || (B) + 2 < G::NHB) DMA_W1_PIECE((B) + 2, (K) * 8 + ROLE * 4 + w4 - 2 * G::CB) }
  // ... and the end of a block: everything but this block's W1 pieces (and NST stores issued behind them) is in; barrier
#define BLK_SYNC(B, ST, NST)                                                                               \
  if (ST || (B) + 2 < G::NHB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R1W + (NST)) : "memory");           \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
  __builtin_amdgcn_s_barrier();
  // the prologue's pieces W1(0), W1(1): piece i * 8 + wavefront
#define DMA_W1_ALL(T)                                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < R1W_ALL; ++i_) {                                                 \
    if (G::KS % 8 == 0 || i_ * 8 + wave < G::KS) DMA_W1_PIECE(T, i_ * 8 + wave)                            \
  }
  for (int i = tid; i < C; i += 512) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  for (int i = tid; i < C; i += 512) { cst[i] = p.ln_w ? p.ln_w[i] : 1.0f; cst[C + i] = p.ln_w ? p.ln_b[i] : 0.0f; }
  const long tile = static_cast<long>(blockIdx.x) * 4 + pair;
  unsigned char* hb_lane = hbuf + pair * 8192 + lane * 32;           // buffer i at + 4096 i: H tile, then (+ 2048) the Hpre tile

  if (wave < 4) {
    // ================================================================ producer: LN, GEMM1, GELU, H -> LDS
    constexpr int ROLE = 0, R1W = G::r1w(ROLE);
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
      DMA_W1_ALL(0)
      DMA_W1_ALL(1)
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
    constexpr int PF = BLK2_PF, NUOP = 4 * 38;
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
  constexpr int ROLE = 1, R1W = G::r1w(ROLE);
  if constexpr (BLK2_PRIO == 2) __builtin_amdgcn_s_setprio(2);
  f32x16 acc2[G::CB];
#pragma unroll
  for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
  DMA_W1_ALL(0)
  DMA_W1_ALL(1)
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Block b:  O += H(b - 2) x W2[b - 2]^T (2 CB MFMAs, b >= 2); between them this wavefront's share of the weight DMA
  constexpr int PFC = BLK2_PFC, DMA_EVERY = 2 * G::CB / NDMA;
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
#undef DMA_W1_ALL
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
    return env ? ((strstr(env, "256") ? 1 : 0) | (strstr(env, "384") ? 2 : 0) | (strstr(env, "192") ? 4 : 0)) : 3;
  }();
  return m;
}
inline bool use_blk2(int C, bool ws) {
  (void)ws;
  const int m = blk2_widths();
  return (C == 256 && (m & 1)) || (C == 384 && (m & 2)) || (BLK2_C192_BUILD && C == 192 && (m & 4));
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


}  // namespace

extern "C" {

int cnx_runtime_switch(int32_t which, int32_t value) {
  switch (which) {
    case CNX_SWITCH_BLK2_WIDTHS: {
      int& m = blk2_widths();
      const int prev = m;
      if (value >= 0) m = value & 7;
      return prev;
    }
    case CNX_SWITCH_DW_SHARED_HALO: return dw_shared_halo_switch(value);
    case CNX_SWITCH_BLK2_BWD_WIDTHS: return blk2b_widths_switch(value);
    case CNX_SWITCH_GEMM_NT_TILE: return gemm_nt_tile_switch(value);
    case CNX_SWITCH_TN_PAIR_RING: return tn_pair_ring_switch(value);
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
#if BLK2_C192_BUILD
    case 192: return use_blk2(192, a.hpre != nullptr) ? launch_blk2_fwd<192>(a, resid_dtype, out_dtype, s) : launch_blk_fwd<192>(a, resid_dtype, out_dtype, s);
#else
    case 192: return launch_blk_fwd<192>(a, resid_dtype, out_dtype, s);
#endif
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

int cnx_block_mlp_fwd_train(const void* u, const float* ln_w, const float* ln_b, float eps, float* mean, float* rstd,
                            const void* Wf, const float* b1, const float* b2, const float* gamma, const void* resid,
                            int resid_dtype, void* out, int out_dtype, void* y2_out, void* hpre_ws, void* h_ws, void* a_rows,
                            int64_t M, int32_t C, void* stream) {
  if (!hpre_ws || !h_ws || !a_rows || !ln_w) return APGD_ERR_NULL;
  if (!cnx_block_mlp_hpre_supported(C)) return APGD_ERR_ARG;
  return block_mlp_fwd_impl(u, ln_w, ln_b, eps, mean, rstd, Wf, b1, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, hpre_ws,
                            M, C, stream, h_ws, a_rows);
}


}  // extern "C"

#if MLP_ABLATE
extern "C" int cnx_dbg_blk_trace(unsigned long long* host, int n_wgs) {
  if (n_wgs > BLK_TRACE_WGS) n_wgs = BLK_TRACE_WGS;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_blk_trace), sizeof(unsigned long long) * BLK_TRACE_SLOTS * n_wgs));
}
#endif
