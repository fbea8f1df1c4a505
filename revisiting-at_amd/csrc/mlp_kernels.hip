// mlp_kernels.hip — fused ConvNeXt MLP for gfx950:  out = x + gamma * (GELU(A W1^T + b1) W2^T + b2)
// (/root/reference/models/convnext.py:42-49: pwconv1 -> GELU -> pwconv2 -> gamma -> residual).
//
// One kernel, two chained bf16 MFMA GEMMs, the [M, 4C] hidden activation never leaves the CU:
//   GEMM1 (per 32-wide slice of the hidden dim):  Ht[h][m] = sum_c W1[h][c] * A[m][c]      (mfma 32x32x16)
//   bias + exact GELU on the fp32 accumulators, rounded to bf16 IN REGISTERS
//   GEMM2:                                         Ot[c][m] += sum_h W2[c][h] * Ht[h][m]
// Computing the TRANSPOSED products makes GEMM1's result layout (lane = column m, 16 rows h per lane)
// directly usable as GEMM2's B operand; the only thing needed is that W2's hidden index is stored in
// the order the accumulator registers enumerate it (`W2p`, permuted within each 32-group on the host):
//   position p = t*16 + half*8 + e   <-   h = (e&3) + 8*(2t + (e>>2)) + 4*half .
// Per workgroup: 4 wavefronts, each owning MB 32-row blocks of M; the 32 x C slice of W1 and the
// C x 32 slice of W2p are staged through LDS (double buffered, padded rows: conflict-free
// ds_read_b128), shared by the 4 wavefronts.  Epilogue: 32x32 fp32 tiles are transposed through LDS
// so that bias, layer scale, residual add and the stores are coalesced along C.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// round-to-nearest-even fp32 pair -> packed bf16 pair: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): one exp, one rcp, 5 fma
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32, 1 ulp
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float z) { return 0.5f * z * (1.0f + erf_as(z * 0.70710678118654752f)); }

template <int C>
struct Cfg {
  static constexpr int MB = (C <= 192) ? 2 : 1;          // 32-row blocks of M per wavefront (measured: 2 beats 1 at C=96/192)
  static constexpr int BM = 4 * 32 * MB;                 // rows of M per workgroup
  static constexpr int ROW1 = C * 2 + 16;                // bytes per padded row of the [32][C] W1 slice
  static constexpr int ROW2 = 64 + 16;                   // bytes per padded row of the [C][32] W2p slice
  static constexpr int T1 = 32 * ROW1;                   // bytes of one W1 slice
  static constexpr int T2 = C * ROW2;                    // bytes of one W2p slice
  static constexpr int STG = 4 * 32 * 33 * 4;            // epilogue staging: per wave 32 x (32+1) fp32
  static constexpr int LDS = 2 * T1 + 2 * T2 + STG;
  static constexpr int KS = C / 16;                      // k-steps of GEMM1
  static constexpr int CB = C / 32;                      // 32-row blocks of the output channels
  static constexpr int NHB = C / 8;                      // 32-wide slices of the hidden dim (4C / 32)
  static constexpr int L1 = (32 * C * 2 / 16 + 255) / 256;   // 16-byte chunks per thread, W1 slice
  static constexpr int L2 = (C * 4 + 255) / 256;              // 16-byte chunks per thread, W2p slice
};

template <int C, typename TX, typename TO>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W1,
                                                         const float* __restrict__ b1, const uint16_t* __restrict__ W2p,
                                                         const float* __restrict__ b2, const float* __restrict__ gamma,
                                                         const TX* __restrict__ resid, TO* __restrict__ out,
                                                         uint16_t* __restrict__ y2_out, long M) {
  using K = Cfg<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* t1 = lds;                       // [2][T1]
  unsigned char* t2 = lds + 2 * K::T1;           // [2][T2]
  float* stg = reinterpret_cast<float*>(lds + 2 * K::T1 + 2 * K::T2);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const long m_wg = static_cast<long>(blockIdx.x) * K::BM;

  // ---- persistent B operand of GEMM1: A^T fragments of this wave's rows (clamped at the tail)
  bf16x8 af[K::MB][K::KS];
#pragma unroll
  for (int mb = 0; mb < K::MB; ++mb) {
    long m = m_wg + (wave * K::MB + mb) * 32 + l32;
    if (m >= M) m = M - 1;
    const uint16_t* ar = A + m * C + half * 8;
#pragma unroll
    for (int ks = 0; ks < K::KS; ++ks) af[mb][ks] = *reinterpret_cast<const bf16x8*>(ar + ks * 16);
  }

  f32x16 acc2[K::MB][K::CB];
#pragma unroll
  for (int mb = 0; mb < K::MB; ++mb)
#pragma unroll
    for (int cb = 0; cb < K::CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mb][cb][r] = 0.f;

  // ---- staging: 16-byte chunks, global -> registers -> LDS (macros, not lambdas: captured arrays
  //      would be demoted to scratch)
  uint4 s1[K::L1], s2[K::L2];
#define G_LOAD(HB)                                                                                              \
  {                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < K::L1; ++i) {                                                         \
      int j = tid + i * 256;                                                                                    \
      j = j < 32 * C / 8 ? j : 32 * C / 8 - 1;   /* unconditional load: keeps s1 in registers */                \
      s1[i] = *reinterpret_cast<const uint4*>(W1 + static_cast<long>(HB) * 32 * C + j * 8);                     \
    }                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < K::L2; ++i) {                                                         \
      int j = tid + i * 256;                                                                                    \
      j = j < C * 4 ? j : C * 4 - 1;                                                                            \
      s2[i] = *reinterpret_cast<const uint4*>(W2p + static_cast<long>(j >> 2) * (4 * C) + (HB) * 32 + (j & 3) * 8); \
    }                                                                                                           \
  }
#define L_STORE(BUF)                                                                                            \
  {                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < K::L1; ++i) {                                                         \
      const int j = tid + i * 256;                                                                              \
      if (j < 32 * C / 8) {                                                                                     \
        const int row = j / (C / 8), c8 = j - row * (C / 8);                                                    \
        *reinterpret_cast<uint4*>(t1 + (BUF) * K::T1 + row * K::ROW1 + c8 * 16) = s1[i];                        \
      }                                                                                                         \
    }                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < K::L2; ++i) {                                                         \
      const int j = tid + i * 256;                                                                              \
      if (j < C * 4) *reinterpret_cast<uint4*>(t2 + (BUF) * K::T2 + (j >> 2) * K::ROW2 + (j & 3) * 16) = s2[i]; \
    }                                                                                                           \
  }

  G_LOAD(0)
  L_STORE(0)
  __syncthreads();

  for (int hb = 0; hb < K::NHB; ++hb) {
    const int buf = hb & 1;
    if (hb + 1 < K::NHB) G_LOAD(hb + 1)                       // in flight while this slice is computed
    __builtin_amdgcn_sched_barrier(0);                        // keep the prefetch loads up here (the scheduler
                                                              // otherwise sinks them next to the LDS stores)

    // ---- GEMM1: Ht[32 h][32 m] per m-block
    f32x16 acc1[K::MB];
#pragma unroll
    for (int mb = 0; mb < K::MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[mb][r] = 0.f;
    const unsigned char* w1r = t1 + buf * K::T1 + l32 * K::ROW1 + half * 16;
#pragma unroll
    for (int ks = 0; ks < K::KS; ++ks) {
      const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w1r + ks * 32);
#pragma unroll
      for (int mb = 0; mb < K::MB; ++mb) acc1[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, af[mb][ks], acc1[mb], 0, 0, 0);
    }
    // ---- bias + GELU on the accumulators; register r <-> h = hb*32 + (r&3) + 8*(r>>2) + 4*half
    bf16x8 hf[K::MB][2];
    {
      float bv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(b1 + hb * 32 + 8 * g + 4 * half);
        bv[4 * g + 0] = b4.x; bv[4 * g + 1] = b4.y; bv[4 * g + 2] = b4.z; bv[4 * g + 3] = b4.w;
      }
#pragma unroll
      for (int mb = 0; mb < K::MB; ++mb) {
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          pk[r >> 1] = pack_bf16(gelu_fast(acc1[mb][r] + bv[r]), gelu_fast(acc1[mb][r + 1] + bv[r + 1]));
        hf[mb][0] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        hf[mb][1] = __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]));
      }
    }
    // ---- GEMM2: Ot[c][m] += W2p slice x Ht
    const unsigned char* w2r = t2 + buf * K::T2 + l32 * K::ROW2 + half * 16;
#pragma unroll
    for (int cb = 0; cb < K::CB; ++cb) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(w2r + cb * 32 * K::ROW2 + t * 32);
#pragma unroll
        for (int mb = 0; mb < K::MB; ++mb)
          acc2[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, hf[mb][t], acc2[mb][cb], 0, 0, 0);
      }
    }
    if (hb + 1 < K::NHB) L_STORE(buf ^ 1)
    __syncthreads();
  }

#undef G_LOAD
#undef L_STORE
  // ---- epilogue: transpose 32x32 tiles through LDS, then coalesced (bias, gamma, residual, store)
  float* ws = stg + wave * 32 * 33;
  const int er = lane >> 3, ec = (lane & 7) * 4;             // read-back: 8 lanes cover 32 channels of a row
#pragma unroll
  for (int mb = 0; mb < K::MB; ++mb) {
    const long m0 = m_wg + (wave * K::MB + mb) * 32;
#pragma unroll
    for (int cb = 0; cb < K::CB; ++cb) {
      // acc2 register r holds Ot[c = (r&3) + 8*(r>>2) + 4*half][m = l32]
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[l32 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc2[mb][cb][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): wave-private region, no barrier needed
      const int c = cb * 32 + ec;
      const float4 bb = *reinterpret_cast<const float4*>(b2 + c);
      float4 gg = make_float4(1.f, 1.f, 1.f, 1.f);
      if (gamma) gg = *reinterpret_cast<const float4*>(gamma + c);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = er + it * 8;
        const long m = m0 + row;
        float4 o;
        o.x = ws[row * 33 + ec + 0] + bb.x; o.y = ws[row * 33 + ec + 1] + bb.y;
        o.z = ws[row * 33 + ec + 2] + bb.z; o.w = ws[row * 33 + ec + 3] + bb.w;
        if (m < M) {
          if (y2_out) {
            uint2 p; p.x = pack_bf16(o.x, o.y); p.y = pack_bf16(o.z, o.w);
            *reinterpret_cast<uint2*>(y2_out + m * C + c) = p;
          }
          o.x *= gg.x; o.y *= gg.y; o.z *= gg.z; o.w *= gg.w;
          if (resid) {
            if constexpr (sizeof(TX) == 4) {
              const float4 xv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resid) + m * C + c);
              o.x += xv.x; o.y += xv.y; o.z += xv.z; o.w += xv.w;
            } else {
              const uint2 xv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(resid) + m * C + c);
              o.x += __uint_as_float(xv.x << 16); o.y += __uint_as_float(xv.x & 0xffff0000u);
              o.z += __uint_as_float(xv.y << 16); o.w += __uint_as_float(xv.y & 0xffff0000u);
            }
          }
          if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + m * C + c) = o;
          } else {
            uint2 p; p.x = pack_bf16(o.x, o.y); p.y = pack_bf16(o.z, o.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + m * C + c) = p;
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  }
}

template <int C>
int launch_mlp_fwd(const void* A, const void* W1, const float* b1, const void* W2p, const float* b2, const float* gamma,
                   const void* resid, int resid_dtype, void* out, int out_dtype, void* y2, long M, hipStream_t s) {
  using K = Cfg<C>;
  const dim3 grid(static_cast<unsigned>((M + K::BM - 1) / K::BM)), block(256);
  const auto* a = static_cast<const uint16_t*>(A);
  const auto* w1 = static_cast<const uint16_t*>(W1);
  const auto* w2 = static_cast<const uint16_t*>(W2p);
  auto* y2p = static_cast<uint16_t*>(y2);
#define MLP_LAUNCH(TX, TO)                                                                                       \
  {                                                                                                              \
    auto kfn = mlp_fwd_kernel<C, TX, TO>;                                                                        \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                K::LDS);                                                                         \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, K::LDS, s, a, w1, b1, w2, b2, gamma, static_cast<const TX*>(resid),      \
                       static_cast<TO*>(out), y2p, M);                                                           \
  }
  if (resid_dtype == APGD_F32 && out_dtype == APGD_F32) MLP_LAUNCH(float, float)
  else if (resid_dtype == APGD_F32) MLP_LAUNCH(float, uint16_t)
  else if (out_dtype == APGD_F32) MLP_LAUNCH(uint16_t, float)
  else MLP_LAUNCH(uint16_t, uint16_t)
#undef MLP_LAUNCH
  return launch_status();
}

}  // namespace

extern "C" {

int cnx_mlp_fwd_supported(int32_t C) { return (C == 96 || C == 192 || C == 384) ? 1 : 0; }

int cnx_mlp_fwd(const void* A, const void* W1, const float* b1, const void* W2p, const float* b2, const float* gamma,
                const void* resid, int resid_dtype, void* out, int out_dtype, void* y2_out, int64_t M, int32_t C,
                void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!A || !W1 || !b1 || !W2p || !b2 || !out) return APGD_ERR_NULL;
  if ((resid_dtype != APGD_F32 && resid_dtype != APGD_BF16) || (out_dtype != APGD_F32 && out_dtype != APGD_BF16))
    return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  switch (C) {
    case 96: return launch_mlp_fwd<96>(A, W1, b1, W2p, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, M, s);
    case 192: return launch_mlp_fwd<192>(A, W1, b1, W2p, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, M, s);
    case 384: return launch_mlp_fwd<384>(A, W1, b1, W2p, b2, gamma, resid, resid_dtype, out, out_dtype, y2_out, M, s);
    default: return APGD_ERR_ARG;
  }
}

}  // extern "C"
