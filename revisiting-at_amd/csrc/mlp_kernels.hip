// mlp_kernels.hip — second-generation fused ConvNeXt block tail for gfx950 (MI355X), forward:
//     out = x + gamma * ( GELU( LN(u) W1^T + b1 ) W2^T + b2 )                 (/root/reference/models/convnext.py:40-49)
//
// What changed against blk_mlp_fwd_kernel (block_kernels.hip), and why (profiles/r01_kernel_microbench.md: the streaming
// phases - 152 us - and the MFMA/GELU phase - 118 us - of the C = 96 kernel ADD UP instead of overlapping; SQ_WAIT_ANY 56-65 %):
//
//   * The packed weights of the stage-0 width (8 C^2 bf16 = 144 KiB at C = 96) FIT the 160 KiB LDS of a CU.  One persistent
//     workgroup per CU loads them ONCE; after that there is no weight stream, no ring, no s_barrier: the four wavefronts of a
//     workgroup are independent and never wait for one another again.
//   * A wavefront owns 64 rows (two 32-row MFMA row groups): every 1 KiB weight fragment read from LDS feeds TWO MFMAs
//     (halves the LDS operand traffic per MFMA), and the two row groups are two independent accumulation chains.
//   * Software pipeline over tiles inside the wavefront: the next tile's rows of u and this tile's residual are requested
//     from HBM BEFORE the hidden loop of this tile runs (they sit in registers: one wavefront per SIMD owns the whole
//     512-register file), the result tile leaves while the next tile computes.  HBM streaming and MFMA work overlap by
//     construction instead of by luck of the co-resident workgroups' phases.
//   * Software pipeline over hidden slices: GELU of slice s (VALU) is independent of GEMM1 of slice s+1 (MFMA) - they are
//     issued back to back so that the matrix pipe and the vector ALU work at the same time inside one wavefront.
//
// Operand conventions, weight packing (cnx_mlp_pack_weights) and numerics are those of block_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "apgd_hip.h"
#include "convnext_hip.h"
#include "mlp_internal.h"

// Timing experiments (APGD_BLK_DBG) are compiled in only with -DMLP_ABLATE=1: a runtime flag test inside the hidden loop
// turns into a branch per use and splits the scheduling region (15 branches, 23 s_nop and 12 s_waitcnt per slice were
// measured in the ISA) - the MFMA / VALU interleave the kernel depends on is gone.
#ifndef MLP_ABLATE
#define MLP_ABLATE 0
#endif
#define DBG(p, bit) (MLP_ABLATE && ((p).dbg & (bit)))

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// GELU(z) = max(z, 0) - 0.5 |z| 2^Q(|z|), Q of degree 5: the exact-erf form to 1.2e-6 (tools/fit_gelu.py); pair -> packed bf16
__device__ __forceinline__ uint32_t gelu2_bf16(float z0, float z1) {
  const f32x2 az = {fabsf(z0), fabsf(z1)};
  f32x2 q = fma2(splat2(-0.00041175442346105595f), az, splat2(0.006678475199902348f));
  q = fma2(q, az, splat2(-0.050879760394516485f));
  q = fma2(q, az, splat2(-0.46094072908550926f));
  q = fma2(q, az, splat2(-1.150400682855232f));
  q = fma2(q, az, splat2(-8.454223479528131e-05f));
  const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const f32x2 pos = {fmaxf(z0, 0.0f), fmaxf(z1, 0.0f)};
  const f32x2 g = fma2(az * e, splat2(-0.5f), pos);
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(g, bf16x2));
}

// Prefetch loads that must stay IN FLIGHT across the hidden loop.  hipcc's own s_waitcnt bookkeeping cannot do that: vmcnt
// retires in order and also counts stores on gfx950, so for loop-carried loads it falls back to `s_waitcnt vmcnt(0)` at the
// first use in every iteration (draining the previous tile's stores as well), and under register pressure it waits for a load
// right after issuing it in order to park the value in the accumulator file.  These loads are therefore hidden from the
// compiler (cdna_hip_programming.md §5.7): inline-asm `global_load_dwordx4` straight into ACCUMULATOR registers (the 256
// AGPRs of a one-wavefront-per-SIMD kernel are otherwise half empty), ONE hand-placed `s_waitcnt vmcnt(0)` at the end of the
// hidden loop - by then the loads, issued ~10 us earlier, and the previous tile's stores have long retired - followed by
// sched_barrier(0) so that no register-only consumer is hoisted above the wait (§5.4 rule 18).
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void prefetch16(f32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void prefetch8(f32x2& dst, const void* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void prefetch_landed() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int C, int RG>
struct GeoR {
  static constexpr int KS = C / 16, CB = C / 32, NHB = C / 8;
  static constexpr int PIECES = KS + 2 * CB;                  // 1 KiB operand fragments per 32-wide hidden slice
  static constexpr int SLICE = PIECES * 1024;
  static constexpr int WBYTES = NHB * SLICE;                  // = 16 C^2 bytes: all of W1 and W2 in fragment order
  static constexpr int CONST_OFF = WBYTES;                    // b1 [4C], b2 [C], gamma [C], ln_w [C], ln_b [C] fp32
  static constexpr int SCR_OFF = CONST_OFF + 32 * C;
  static constexpr int SCR_WAVE = 8 * C * 4;                  // 8 rows x C fp32: one transposition pass of the epilogue
  static constexpr int LDS = SCR_OFF + 4 * SCR_WAVE;
  static constexpr int ROWS = 32 * RG;                        // rows per wavefront tile
  static constexpr int NCH = 8 * C / 4 / 64;                  // float4 chunks per lane in one 8-row pass
  static_assert(LDS <= 160 * 1024, "weights + scratch must fit the CU's LDS");
  static_assert((8 * C / 4) % 64 == 0, "8-row passes must tile into whole wavefront float4 sweeps");
  static_assert((NHB * PIECES) % 4 == 0, "pieces divide over 4 wavefronts");
};

// one hidden slice s:   hf = GELU(acc_in)   |   acc_out = b1[s+1] + W1[s+1] a^T   (NEXT)   |   acc2 += hf W2[s]^T
// The W1 fragments of slice s+1 and the W2 fragments of slice s form one stream of 1 KiB LDS reads, PF fragments ahead of
// the MFMAs that consume them; each fragment feeds RG MFMAs (one per row group).
template <int C, int RG, bool NEXT>
__device__ __forceinline__ void slice_step(int dbg, const unsigned char* w_lane, const float* b1s, int s, int half,
                                           const bf16x8 (&af)[RG][C / 16], const f32x16 (&acc_in)[RG], f32x16 (&acc_out)[RG],
                                           f32x16 (&acc2)[RG][C / 32]) {
  using G = GeoR<C, RG>;
  constexpr int KS = G::KS, CB = G::CB;
  constexpr int N1 = NEXT ? KS : 0, NF = N1 + 2 * CB, PF = 4;
  const unsigned char* w_cur = w_lane + static_cast<long>(s) * G::SLICE + KS * 1024;          // W2 fragments of slice s
  const unsigned char* w_nxt = w_lane + static_cast<long>(s + 1) * G::SLICE;                  // W1 fragments of slice s + 1
  auto frag = [&](int i) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(i < N1 ? w_nxt + i * 1024 : w_cur + (i - N1) * 1024);
  };
  bf16x8 fr[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) fr[i] = frag(i);
  if constexpr (NEXT) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + (s + 1) * 32 + 8 * g4 + 4 * half);
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        acc_out[g][4 * g4 + 0] = b4.x; acc_out[g][4 * g4 + 1] = b4.y; acc_out[g][4 * g4 + 2] = b4.z; acc_out[g][4 * g4 + 3] = b4.w;
      }
    }
  }
  // GELU of slice s -> bf16 A-operand fragments of GEMM2 (accumulator layout of H^T == A layout of H, k permuted)
  bf16x8 hf[RG][2];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    uint32_t pk[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2)
      pk[r >> 1] = (MLP_ABLATE && (dbg & 1)) ? pack_bf16(acc_in[g][r], acc_in[g][r + 1]) : gelu2_bf16(acc_in[g][r], acc_in[g][r + 1]);   // dbg 1: timing experiment, no GELU
    hf[g][0] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    hf[g][1] = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if (i < N1) {
#pragma unroll
      for (int g = 0; g < RG; ++g) acc_out[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[g][i], acc_out[g], 0, 0, 0);
    } else {
      const int j = i - N1;                                   // fragment order in the slice is (t, cb): j = t * CB + cb
#pragma unroll
      for (int g = 0; g < RG; ++g)
        acc2[g][j % CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[g][j / CB], fr[i % PF], acc2[g][j % CB], 0, 0, 0);
    }
    if (i + PF < NF) fr[i % PF] = frag(i + PF);
  }
}

template <int C, int RG, typename TX, typename TO>
__global__ __launch_bounds__(256, 1) void mlp2_fwd_res_kernel(const BlkFwdArgs p) {
  using G = GeoR<C, RG>;
  constexpr int KS = G::KS, CB = G::CB, NHB = G::NHB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* b1s = reinterpret_cast<float*>(lds + G::CONST_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;

  // ---- once per workgroup: all weight fragments L2 -> LDS (async DMA, 1 KiB per instruction), constants
  {
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wf) + lane * 16;
#pragma unroll 4
    for (int i = 0; i < NHB * G::PIECES / 4; ++i) {
      const int piece = i * 4 + wave;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc + static_cast<long>(piece) * 1024), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
    }
    for (int i = tid; i < C; i += 256) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    for (int i = tid; i < C; i += 256) {
      b1s[4 * C + i] = p.b2[i];
      b1s[5 * C + i] = p.gamma ? p.gamma[i] : 1.0f;
      b1s[6 * C + i] = p.ln_w ? p.ln_w[i] : 1.0f;
      b1s[7 * C + i] = p.ln_w ? p.ln_b[i] : 0.0f;
    }
  }

  const long n_tiles = (p.M + G::ROWS - 1) / G::ROWS;
  long tile = static_cast<long>(blockIdx.x) * 4 + wave;
  const long tstride = static_cast<long>(gridDim.x) * 4;

  // raw rows of u for a tile: lane (row = l32 of row group g, half) holds channels half * C/2 + ks * 8 + (0..7)
  f32x4 raw[RG][KS];
  auto load_u = [&](long t) {
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      long row = t * G::ROWS + g * 32 + l32;
      if (row >= p.M) row = p.M - 1;
      const unsigned char* up = reinterpret_cast<const unsigned char*>(p.u + row * C + half * (C / 2));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) prefetch16(raw[g][ks], up + ks * 16);
    }
  };
  if (tile < n_tiles) load_u(tile);

  prefetch_landed();                                          // the weight DMA and the first tile's rows
  __syncthreads();                                            // weights and constants are in; last barrier of the kernel

  const unsigned char* w_lane = lds + lane * 16;
  float* scr = reinterpret_cast<float*>(lds + G::SCR_OFF + wave * G::SCR_WAVE);
  const float4* b2v = reinterpret_cast<const float4*>(b1s + 4 * C);
  const float4* gav = reinterpret_cast<const float4*>(b1s + 5 * C);
  constexpr int C4 = C / 4;
  const TX* resid = static_cast<const TX*>(p.resid);
  TO* out = static_cast<TO*>(p.out);
  const long e_end = p.M * C;

  for (; tile < n_tiles; tile += tstride) {
    const long m0 = tile * G::ROWS;
    // ---- LayerNorm of this tile's rows -> GEMM1 B-operand fragments
    bf16x8 af[RG][KS];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const long row = m0 + g * 32 + l32;
      if (p.ln_w) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint32_t w[4] = {__float_as_uint(raw[g][ks][0]), __float_as_uint(raw[g][ks][1]), __float_as_uint(raw[g][ks][2]), __float_as_uint(raw[g][ks][3])};
#pragma unroll
          for (int j = 0; j < 4; ++j) s += bf16_lo(w[j]) + bf16_hi(w[j]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint32_t w[4] = {__float_as_uint(raw[g][ks][0]), __float_as_uint(raw[g][ks][1]), __float_as_uint(raw[g][ks][2]), __float_as_uint(raw[g][ks][3])};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = bf16_lo(w[j]) - mean, b = bf16_hi(w[j]) - mean;
            ss = fmaf(a, a, ss);
            ss = fmaf(b, b, ss);
          }
        }
        ss += __shfl_xor(ss, 32, 64);
        const float rstd = rsqrtf(ss * (1.0f / C) + p.eps);
        if (p.mean && half == 0 && row < p.M) { p.mean[row] = mean; p.rstd[row] = rstd; }
        const float4* lw = reinterpret_cast<const float4*>(b1s + 6 * C + half * (C / 2));
        const float4* lb = reinterpret_cast<const float4*>(b1s + 7 * C + half * (C / 2));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint32_t w[4] = {__float_as_uint(raw[g][ks][0]), __float_as_uint(raw[g][ks][1]), __float_as_uint(raw[g][ks][2]), __float_as_uint(raw[g][ks][3])};
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
          af[g][ks] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[g][ks] = __builtin_bit_cast(bf16x8, raw[g][ks]);
      }
    }
    // ---- requests that ride under the hidden loop: the next tile's rows, this tile's residual (epilogue chunk order)
    const long nt = tile + tstride;
    if (nt < n_tiles && !DBG(p, 8)) load_u(nt);             // dbg 8: timing experiment, no HBM loads after the first tile
    // (addresses of rows past the end are clamped: whatever is loaded there is never stored)
    using RT = typename std::conditional<sizeof(TX) == 4, f32x4, f32x2>::type;
    RT res[RG][4][G::NCH];
    const bool has_res = resid != nullptr && !DBG(p, 8);
    if (has_res) {
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < G::NCH; ++j) {
            long e = (m0 + g * 32 + q * 8) * C + (j * 64 + lane) * 4;
            if (e >= e_end) e = 0;
            if constexpr (sizeof(TX) == 4) prefetch16(res[g][q][j], reinterpret_cast<const float*>(resid) + e);
            else prefetch8(res[g][q][j], reinterpret_cast<const uint16_t*>(resid) + e);
          }
    }

    // ---- hidden loop
    f32x16 acc2[RG][CB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[g][cb][r] = 0.f;
    f32x16 accA[RG], accB[RG];
    {                                                         // GEMM1 of slice 0
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 b4 = *reinterpret_cast<const float4*>(b1s + 8 * g4 + 4 * half);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          accA[g][4 * g4 + 0] = b4.x; accA[g][4 * g4 + 1] = b4.y; accA[g][4 * g4 + 2] = b4.z; accA[g][4 * g4 + 3] = b4.w;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 f = *reinterpret_cast<const bf16x8*>(w_lane + ks * 1024);
#pragma unroll
        for (int g = 0; g < RG; ++g) accA[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, af[g][ks], accA[g], 0, 0, 0);
      }
    }
    static_assert(NHB % 2 == 0, "the slice loop is unrolled by two (accumulator ping-pong)");
    const int n_main = DBG(p, 4) ? 0 : NHB - 2;             // dbg 4: timing experiment, two of the twelve slices only
#pragma unroll 1
    for (int s = 0; s < n_main; s += 2) {
      slice_step<C, RG, true>(p.dbg, w_lane, b1s, s, half, af, accA, accB, acc2);
      slice_step<C, RG, true>(p.dbg, w_lane, b1s, s + 1, half, af, accB, accA, acc2);
    }
    slice_step<C, RG, true>(p.dbg, w_lane, b1s, NHB - 2, half, af, accA, accB, acc2);
    slice_step<C, RG, false>(p.dbg, w_lane, b1s, NHB - 1, half, af, accB, accA, acc2);

    prefetch_landed();                                        // next tile's rows + this tile's residual (and the previous tile's stores)
    // ---- epilogue: acc2[g][cb][r] = O[m0 + 32 g + (r&3) + 8 (r>>2) + 4 half][cb*32 + l32].  Eight rows at a time go through this
    //      wavefront's 8 x C fp32 scratch and leave as whole rows: 16 bytes per lane for the scratch read, the residual (already
    //      in registers) and the store.  LDS operations of one wavefront execute in order: no barrier, only compiler fences.
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int i = 0; i < 4; ++i) scr[(i + 4 * half) * C + cb * 32 + l32] = acc2[g][cb][4 * q + i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < G::NCH; ++j) {
          const int idx = j * 64 + lane;
          const long e = (m0 + g * 32 + q * 8) * C + idx * 4;
          const int c4 = idx % C4;
          const float4 o = reinterpret_cast<const float4*>(scr)[idx];
          const float4 bb = b2v[c4], gg = gav[c4];
          const float y0 = o.x + bb.x, y1 = o.y + bb.y, y2v = o.z + bb.z, y3 = o.w + bb.w;
          if (e < e_end && !DBG(p, 2)) {                     // dbg 2: timing experiment, no stores
            if (p.y2) *reinterpret_cast<uint2*>(p.y2 + e) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2v, y3));
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_res) {
              if constexpr (sizeof(TX) == 4) {
                xv = make_float4(res[g][q][j][0], res[g][q][j][1], res[g][q][j][2], res[g][q][j][3]);
              } else {
                const uint32_t w0 = __float_as_uint(res[g][q][j][0]), w1 = __float_as_uint(res[g][q][j][1]);
                xv = make_float4(bf16_lo(w0), bf16_hi(w0), bf16_lo(w1), bf16_hi(w1));
              }
            }
            const float o0 = fmaf(y0, gg.x, xv.x), o1 = fmaf(y1, gg.y, xv.y);
            const float o2 = fmaf(y2v, gg.z, xv.z), o3 = fmaf(y3, gg.w, xv.w);
            if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + e) = make_float4(o0, o1, o2, o3);
            else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + e) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
          }
        }
      }
  }
}

// =====================================================================================================================
// Third form (the one the product uses at C = 96): the same LDS-resident weights, but NW = 12 (or 8) INDEPENDENT wavefronts
// per workgroup - three (two) per SIMD - each walking its own 32 x RG-row tiles, plain compiler-scheduled loads.
// Measured on MI355X (gpurun_out/r2_mlp_bench2.log): with one wavefront per SIMD (the form above) the hidden loop is bound by
// the LATENCY of the GELU's dependent VALU chains - 2000 cycles per 12-MFMA slice against 960 in the first-generation kernel
// with three wavefronts per SIMD - so the freed barrier time was lost again.  Here nothing ties the wavefronts of a CU
// together (no ring, no s_barrier after the weights have landed): they drift apart, and while some are in their HBM phases
// (rows in, residual in, result out) the others keep the matrix pipe and the VALU busy.
//   * the residual tile and the next tile's rows are TOUCHED (one dword per 128-byte line) before the hidden loop, so that
//     the epilogue's and the next prologue's loads find them in L2 / Infinity Cache; the hidden loop itself has no VMEM;
//   * the epilogue requests a whole row group's residual before its first pass (it was one dependent HBM round trip per
//     2-row pass: 94 of 324 us);
//   * GELU(z) = 0.5 z + |z| (0.5 - 0.5 E), E = erfc(|z|/sqrt 2): 13 VALU per pair instead of 17 (no max, no canonicalising
//     max in front of it) - with every wave64 VALU instruction costing 4 cycles here, the instruction count IS the time.
__device__ __forceinline__ uint32_t gelu2b_bf16(float z0, float z1) {
  const f32x2 z = {z0, z1};
  const f32x2 az = {fabsf(z0), fabsf(z1)};
  f32x2 q = fma2(splat2(-0.00041175442346105595f), az, splat2(0.006678475199902348f));
  q = fma2(q, az, splat2(-0.050879760394516485f));
  q = fma2(q, az, splat2(-0.46094072908550926f));
  q = fma2(q, az, splat2(-1.150400682855232f));
  q = fma2(q, az, splat2(-8.454223479528131e-05f));
  const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const f32x2 w = fma2(e, splat2(-0.5f), splat2(0.5f));
  const f32x2 g = fma2(az, w, z * splat2(0.5f));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(g, bf16x2));
}

template <int C, int NW, int RG>
struct Geo3 {
  using G = GeoR<C, 1>;
  static constexpr int SCR_OFF = G::CONST_OFF + 32 * C;       // after b1, b2, gamma, ln_w, ln_b
  static constexpr int SCR_WAVE = 2 * C * 4;                  // 2 rows x C fp32 per wavefront
  static constexpr int LDS = SCR_OFF + NW * SCR_WAVE;
  static constexpr int ROWS = 32 * RG;
  static_assert(LDS <= 160 * 1024, "weights + constants + scratch must fit the CU's LDS");
  static_assert((G::NHB * G::PIECES) % NW == 0, "pieces divide over the wavefronts");
  static_assert(2 * (C / 4) <= 64, "a 2-row pass is one float4 per lane");
};

template <int C, int NW, int RG, typename TX, typename TO>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp3_fwd_res_kernel(const BlkFwdArgs p) {
  using G = GeoR<C, 1>;
  using G3 = Geo3<C, NW, RG>;
  constexpr int KS = G::KS, CB = G::CB, NHB = G::NHB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* b1s = reinterpret_cast<float*>(lds + G::CONST_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  {
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wf) + lane * 16;
#pragma unroll 4
    for (int i = 0; i < NHB * G::PIECES / NW; ++i) {
      const int piece = i * NW + wave;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc + static_cast<long>(piece) * 1024), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
    }
    for (int i = tid; i < C; i += NW * 64) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    for (int i = tid; i < C; i += NW * 64) {
      b1s[4 * C + i] = p.b2[i];
      b1s[5 * C + i] = p.gamma ? p.gamma[i] : 1.0f;
      b1s[6 * C + i] = p.ln_w ? p.ln_w[i] : 1.0f;
      b1s[7 * C + i] = p.ln_w ? p.ln_b[i] : 0.0f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                            // the only barrier of the kernel
  // Symmetry breaking.  The NW / 4 wavefronts that share a SIMD start together and do identical work, so without help they
  // stay in phase for the whole kernel: all in their HBM phases at once (SIMD idle), then all in the hidden loop at once.
  // Wavefront slot k of a SIMD (waves k*4 .. k*4+3) therefore starts `stagger` x k sleep quanta late, once; the persistent
  // tile loop keeps the offset.  (p.dbg >> 8 = quanta of s_sleep 127 ~ 4 us each; 0 = off.)
  {
    const int quanta = (p.dbg >> 8) * (wave >> 2);
    for (int i = 0; i < quanta; ++i) __builtin_amdgcn_s_sleep(127);
  }

  const unsigned char* w_lane = lds + lane * 16;
  float* scr = reinterpret_cast<float*>(lds + G3::SCR_OFF + wave * G3::SCR_WAVE);
  const float4* b2v = reinterpret_cast<const float4*>(b1s + 4 * C);
  const float4* gav = reinterpret_cast<const float4*>(b1s + 5 * C);
  constexpr int C4 = C / 4;
  const TX* resid = static_cast<const TX*>(p.resid);
  TO* out = static_cast<TO*>(p.out);
  const long n_tiles = (p.M + G3::ROWS - 1) / G3::ROWS;
  const long tstride = static_cast<long>(gridDim.x) * NW;
  const int rr = lane / C4, c4 = lane - rr * C4;               // epilogue: row of a 2-row pass, float4 column of this lane
  const bool ep_lane = lane < 2 * C4;

  for (long tile = static_cast<long>(blockIdx.x) * NW + wave; tile < n_tiles; tile += tstride) {
    const long m0 = tile * G3::ROWS;
    bf16x8 af[RG][KS];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      long row = m0 + g * 32 + l32;
      const bool row_ok = row < p.M;
      if (!row_ok) row = p.M - 1;
      uint4 raw[KS];
      const uint4* up = reinterpret_cast<const uint4*>(p.u + row * C + half * (C / 2));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) raw[ks] = up[ks];
      if (p.ln_w) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) s += bf16_lo(w[j]) + bf16_hi(w[j]);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
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
        const float4* lw = reinterpret_cast<const float4*>(b1s + 6 * C + half * (C / 2));
        const float4* lb = reinterpret_cast<const float4*>(b1s + 7 * C + half * (C / 2));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint32_t w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
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
          af[g][ks] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[g][ks] = __builtin_bit_cast(bf16x8, raw[ks]);
      }
    }
    // ---- touch (one dword per 128-byte line) what this wavefront reads next from HBM: the residual tile of this tile and the
    //      rows of its next tile.  The values are dropped; the lines are in L2 / Infinity Cache when the real loads come.
    constexpr int kResLines = (G3::ROWS * C * static_cast<int>(sizeof(TX)) + 127) / 128;
    constexpr int kULines = (G3::ROWS * C * 2 + 127) / 128;
    constexpr int kPf = (kResLines + 63) / 64 + (kULines + 63) / 64;
    uint32_t pf[kPf];
    {
      int k = 0;
      const long lim = (p.M - m0) * C * static_cast<long>(sizeof(TX));
      const unsigned char* rb = reinterpret_cast<const unsigned char*>(resid) + m0 * C * static_cast<long>(sizeof(TX));
#pragma unroll
      for (int i = 0; i < (kResLines + 63) / 64; ++i, ++k) {
        long off = (static_cast<long>(i) * 64 + lane) * 128;
        if (off >= lim || off >= static_cast<long>(kResLines) * 128) off = 0;
        pf[k] = (resid && !DBG(p, 8)) ? *reinterpret_cast<const uint32_t*>(rb + off) : 0u;
      }
      const long nm0 = (tile + tstride) * G3::ROWS;
      const long ulim = (p.M - nm0) * C * 2;
      const unsigned char* ub = reinterpret_cast<const unsigned char*>(p.u) + nm0 * C * 2;
#pragma unroll
      for (int i = 0; i < (kULines + 63) / 64; ++i, ++k) {
        long off = (static_cast<long>(i) * 64 + lane) * 128;
        if (off >= static_cast<long>(kULines) * 128) off = 0;
        pf[k] = (nm0 < p.M && off < ulim && !DBG(p, 8)) ? *reinterpret_cast<const uint32_t*>(ub + off) : 0u;
      }
    }

    f32x16 acc2[RG][CB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[g][cb][r] = 0.f;
    const int n_sl = DBG(p, 4) ? 2 : NHB;                     // dbg 4: timing experiment, two of the twelve slices only
#pragma unroll 1
    for (int s = 0; s < n_sl; ++s) {
      const unsigned char* sl = w_lane + static_cast<long>(s) * G::SLICE;
      constexpr int NF = KS + 2 * CB, PF = (NW >= 16 ? 2 : 4);
      bf16x8 fr[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(sl + i * 1024);
      f32x16 acc1[RG];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 b4 = *reinterpret_cast<const float4*>(b1s + s * 32 + 8 * g4 + 4 * half);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          acc1[g][4 * g4 + 0] = b4.x; acc1[g][4 * g4 + 1] = b4.y; acc1[g][4 * g4 + 2] = b4.z; acc1[g][4 * g4 + 3] = b4.w;
        }
      }
#pragma unroll
      for (int i = 0; i < KS; ++i) {                          // (one chain per row group: the co-resident wavefronts fill the gaps)
#pragma unroll
        for (int g = 0; g < RG; ++g) acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % PF], af[g][i], acc1[g], 0, 0, 0);
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      }
      bf16x8 hf[RG][2];
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          pk[r >> 1] = DBG(p, 1) ? pack_bf16(acc1[g][r], acc1[g][r + 1]) : gelu2b_bf16(acc1[g][r], acc1[g][r + 1]);   // dbg 1: no GELU
        hf[g][0] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        hf[g][1] = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
      }
#pragma unroll
      for (int j = 0; j < 2 * CB; ++j) {
        const int i = KS + j;
#pragma unroll
        for (int g = 0; g < RG; ++g)
          acc2[g][j % CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[g][j / CB], fr[i % PF], acc2[g][j % CB], 0, 0, 0);
        if (i + PF < NF) fr[i % PF] = *reinterpret_cast<const bf16x8*>(sl + (i + PF) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < kPf; ++i) asm volatile("" ::"v"(pf[i]));          // the touch loads were issued (and are complete) here

    // ---- epilogue: EB 2-row passes at a time - their residual requests first, then two rows (r and r + 4 of an 8-row group)
    //      per pass through 2 x C floats of scratch; lanes 0 .. 2 C/4 - 1 move 16 bytes each of the residual and of the result
    const int rows_left = static_cast<int>(p.M - m0 < G3::ROWS ? p.M - m0 : G3::ROWS);
    const long ebase = (m0 + 4 * rr) * C + c4 * 4;            // this lane's element offset in pass r = 0 of row group 0
    constexpr int EB = NW >= 16 ? 4 : 8;                      // passes per batch of residual requests (registers)
#pragma unroll
    for (int gh = 0; gh < (16 / EB) * RG; ++gh) {             // (row group, batch of its 16 passes)
      const int g = gh / (16 / EB), r0 = (gh % (16 / EB)) * EB;
      float4 xv[EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const int r = r0 + k, ro = g * 32 + (r & 3) + 8 * (r >> 2);      // row offset of lane row 0 of this pass
        xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (resid && ep_lane && ro + 4 * rr < rows_left && !DBG(p, 2)) {
          if constexpr (sizeof(TX) == 4) {
            xv[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resid) + ebase + ro * C);
          } else {
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(resid) + ebase + ro * C);
            xv[k] = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
          }
        }
      }
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const int r = r0 + k, ro = g * 32 + (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) scr[half * C + cb * 32 + l32] = acc2[g][cb][r];
        __builtin_amdgcn_wave_barrier();
        if (ep_lane && ro + 4 * rr < rows_left && !DBG(p, 2)) {
          const long e = ebase + ro * C;
          const float4 o = reinterpret_cast<const float4*>(scr)[lane];
          const float4 bb = b2v[c4], gg = gav[c4];
          const float y0 = o.x + bb.x, y1 = o.y + bb.y, y2v = o.z + bb.z, y3 = o.w + bb.w;
          if (p.y2) *reinterpret_cast<uint2*>(p.y2 + e) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2v, y3));
          const float o0 = fmaf(y0, gg.x, xv[k].x), o1 = fmaf(y1, gg.y, xv[k].y);
          const float o2 = fmaf(y2v, gg.z, xv[k].z), o3 = fmaf(y3, gg.w, xv[k].w);
          if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + e) = make_float4(o0, o1, o2, o3);
          else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + e) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
        }
      }
    }
  }
}

template <int C, int NW, int RG>
int launch_res3(const BlkFwdArgs& a, int resid_dtype, int out_dtype, hipStream_t s) {
  using G3 = Geo3<C, NW, RG>;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  const long n_tiles = (a.M + G3::ROWS - 1) / G3::ROWS;
  long nb = (n_tiles + NW - 1) / NW;
  if (nb > n_cu) nb = n_cu;
  const dim3 grid(static_cast<unsigned>(nb)), block(NW * 64);
#define MLP3_LAUNCH(TX, TO)                                                                                      \
  {                                                                                                              \
    auto kfn = mlp3_fwd_res_kernel<C, NW, RG, TX, TO>;                                                           \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                G3::LDS);                                                                        \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, G3::LDS, s, a);                                                         \
  }
  if (resid_dtype == APGD_F32 && out_dtype == APGD_F32) MLP3_LAUNCH(float, float)
  else if (resid_dtype == APGD_F32) MLP3_LAUNCH(float, uint16_t)
  else if (out_dtype == APGD_F32) MLP3_LAUNCH(uint16_t, float)
  else MLP3_LAUNCH(uint16_t, uint16_t)
#undef MLP3_LAUNCH
  return static_cast<int>(hipGetLastError());
}

template <int C, int RG>
int launch_res(const BlkFwdArgs& a, int resid_dtype, int out_dtype, hipStream_t s) {
  using G = GeoR<C, RG>;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  const long n_tiles = (a.M + G::ROWS - 1) / G::ROWS;
  long nb = (n_tiles + 3) / 4;
  if (nb > n_cu) nb = n_cu;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
#define MLP2_LAUNCH(TX, TO)                                                                                      \
  {                                                                                                              \
    auto kfn = mlp2_fwd_res_kernel<C, RG, TX, TO>;                                                               \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                G::LDS);                                                                         \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, G::LDS, s, a);                                                          \
  }
  if (resid_dtype == APGD_F32 && out_dtype == APGD_F32) MLP2_LAUNCH(float, float)
  else if (resid_dtype == APGD_F32) MLP2_LAUNCH(float, uint16_t)
  else if (out_dtype == APGD_F32) MLP2_LAUNCH(uint16_t, float)
  else MLP2_LAUNCH(uint16_t, uint16_t)
#undef MLP2_LAUNCH
  return static_cast<int>(hipGetLastError());
}

}  // namespace

int mlp2_fwd_launch(const BlkFwdArgs& a, int C, int resid_dtype, int out_dtype, hipStream_t s) {
  // APGD_MLP2_RG (tuning experiments only): 0 = twelve independent wavefronts per CU, 32 rows each (mlp3, the default);
  // 8 = eight wavefronts, 64 rows each; 1 / 2 = the one-wavefront-per-SIMD form with register prefetch (mlp2)
  static const int rg = getenv("APGD_MLP2_RG") ? atoi(getenv("APGD_MLP2_RG")) : 0;
  if (C != 96) return -100;
  if (rg == 1) return launch_res<96, 1>(a, resid_dtype, out_dtype, s);
  if (rg == 2) return launch_res<96, 2>(a, resid_dtype, out_dtype, s);
  if (rg == 8) return launch_res3<96, 8, 2>(a, resid_dtype, out_dtype, s);
  if (rg == 16) return launch_res3<96, 16, 1>(a, resid_dtype, out_dtype, s);
  return launch_res3<96, 12, 1>(a, resid_dtype, out_dtype, s);
}
