// attn_kernels.hip — fused multi-head softmax attention for the ViT family on gfx950.
//   timm 0.8 `Attention.forward` (called from /root/reference/utils_architecture.py:272-301 models):
//       qkv = Linear(C, 3C)(x).reshape(B, N, 3, h, d).permute(2, 0, 3, 1, 4);  a = softmax(q k^T * d^-1/2);  out = a v
//   Sequence lengths here are tiny (N = 197 @224, 401 @320): one workgroup owns one (batch, head) pair, K and V of the
//   head live in LDS for the whole kernel and the N x N score matrix never exists in memory.
//
//   Layout trick (the same one the fused MLP uses): compute the TRANSPOSED scores  S^T[k][q] = K Q^T  with
//   mfma 32x32x16 (A = K rows from LDS, B = Q rows from registers).  In the accumulator a lane then holds ONE query q
//   and its registers enumerate the keys, so the softmax statistics (max, sum) of a row are in-lane reductions plus one
//   exchange between the two half-waves, and the probabilities - rounded to bf16 in registers - are already an A
//   operand (lane = q, k in accumulator order) for  O[q][d] = P V.  V is therefore stored in LDS as B fragments with
//   the SAME key order ("lane = d, registers = k"): a transposing scatter at staging time, once per workgroup.
//   K is stored as A fragments (lane = key, 8 consecutive d) - every ds_read_b128 is linear and conflict-free.
//   O comes out with lane = d: for a fixed register the 32 lanes of a half-wave store 32 consecutive channels of a row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

constexpr int kD = 64;            // head dimension (ViT-S/B/M: 384/6, 768/12, 512/8)
constexpr int kKS = kD / 16;      // k-steps of Q K^T
constexpr int kDB = kD / 32;      // 32-wide blocks of d

// NKB = number of 32-key blocks (7 for N <= 224, 13 for N <= 416)
template <int NKB>
__global__ __launch_bounds__(256, (NKB <= 7 ? 2 : 1)) void attn_fwd_kernel(const uint16_t* __restrict__ qkv,
                                                                          uint16_t* __restrict__ out, float* __restrict__ lse,
                                                                          int N, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* kimg = lds;                                   // [NKB][kKS][64 lanes][16 B]
  uint16_t* vimg = reinterpret_cast<uint16_t*>(lds + NKB * kKS * 1024);   // [NKB][2 t][kDB][64 lanes][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long tok_stride = 3L * H * kD;                         // elements between consecutive tokens
  const uint16_t* qb = qkv + static_cast<long>(b) * N * tok_stride + h * kD;
  const uint16_t* kb_ = qb + static_cast<long>(H) * kD;
  const uint16_t* vb = qb + 2L * H * kD;

  // ---- stage K (A fragments) and V (transposed B fragments); keys >= N are zero
  for (int c = tid; c < NKB * 32 * (kD / 8); c += 256) {
    const int k = c >> 3, d8 = c & 7;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (k < N) {
      kv = *reinterpret_cast<const uint4*>(kb_ + k * tok_stride + d8 * 8);
      vv = *reinterpret_cast<const uint4*>(vb + k * tok_stride + d8 * 8);
    }
    const int kblk = k >> 5, kk = k & 31;
    *reinterpret_cast<uint4*>(kimg + (((kblk * kKS + (d8 >> 1)) * 64) + (d8 & 1) * 32 + kk) * 16) = kv;
    const int vh = (kk >> 2) & 1, t = kk >> 4, e = (kk & 3) + 4 * ((kk >> 3) & 1);
    const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = d8 * 8 + j, db = d >> 5, dl = d & 31;
      const uint16_t val = static_cast<uint16_t>((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
      vimg[((((kblk * 2 + t) * kDB + db) * 64) + vh * 32 + dl) * 8 + e] = val;
    }
  }
  __syncthreads();

  const int nqb = (N + 31) / 32;
  const unsigned char* kfr = kimg + lane * 16;
  const uint16_t* vfr = vimg + lane * 8;
  for (int qblk = wave; qblk < nqb; qblk += 4) {
    // opaque per q-block: otherwise the (loop-invariant) 56 operand fragments are hoisted out of the loop into 224
    // registers and the kernel spills
    asm volatile("" : "+v"(kfr), "+v"(vfr));
    int q = qblk * 32 + l32;
    const bool q_ok = q < N;
    if (!q_ok) q = N - 1;
    bf16x8 qf[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qb + q * tok_stride + ks * 16 + half * 8);

    // ---- S^T = K Q^T : lane = q, register r of block kblk <-> key kblk*32 + (r&3) + 8*(r>>2) + 4*half
    f32x16 s[NKB];
#pragma unroll
    for (int kblk = 0; kblk < NKB; ++kblk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kblk][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kfr + (kblk * kKS + ks) * 1024);
        s[kblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], s[kblk], 0, 0, 0);
      }
    }
    // ---- softmax over keys (in-lane + one exchange with the other half-wave); keys >= N masked out
    float mx = -INFINITY;
#pragma unroll
    for (int kblk = 0; kblk < NKB; ++kblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= N) s[kblk][r] = -INFINITY;
        mx = fmaxf(mx, s[kblk][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mb = mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kblk = 0; kblk < NKB; ++kblk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[kblk][r], scale_log2e, -mb));
        s[kblk][r] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lse && q_ok && half == 0)
      lse[(static_cast<long>(b) * H + h) * N + q] = (mb + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f;   // natural log

    // ---- O = P V : A = P (lane = q, k in accumulator order), B = V fragments
    f32x16 o[kDB];
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kblk = 0; kblk < NKB; ++kblk) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack_bf16(s[kblk][t * 8 + 2 * j] * inv, s[kblk][t * 8 + 2 * j + 1] * inv);
        const bf16x8 pa = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
          const bf16x8 vbf = *reinterpret_cast<const bf16x8*>(vfr + ((kblk * 2 + t) * kDB + db) * 512);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vbf, o[db], 0, 0, 0);
        }
      }
    }
    // ---- store: o[db][r] = O[qblk*32 + (r&3) + 8*(r>>2) + 4*half][db*32 + l32]
    uint16_t* ob = out + static_cast<long>(b) * N * H * kD + h * kD;
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (qr < N) ob[static_cast<long>(qr) * H * kD + db * 32 + l32] = static_cast<uint16_t>(pack_bf16(o[db][r], 0.f));
      }
  }
}

// =====================================================================================================================
// Backward.  With P = exp(S*scale - lse) (lse saved by the forward) no second softmax pass is needed and both kernels
// stream 32x32 blocks of the score matrix through 32 accumulator registers:
//     D_q   = sum_d dO[q][d] O[q][d]
//     dV    = P^T dO            dP = dO V^T           dS = P * (dP - D_q) * scale
//     dQ    = dS K              dK = dS^T Q
// A contraction over keys needs "lane = query, registers = keys" (the S^T orientation), one over queries needs
// "lane = key, registers = queries" (the S orientation); the MFMA accumulator gives whichever is asked for by swapping
// the operands, so there are two kernels:
//   attn_bwd_dq_kernel  : wavefront = 32 queries, loops over key blocks:   S^T, dP^T -> dS -> dQ += dS K      (+ writes D_q)
//   attn_bwd_dkv_kernel : wavefront = 32 keys,    loops over query blocks: S, dP -> P, dS -> dV += P^T dO, dK += dS^T Q
// Operand images in LDS are the two kinds of the forward: "row" images (lane = row, 8 consecutive d: A/B operands of a
// contraction over d) and "transposed" images (lane = d, registers = rows in accumulator order: B operands of a
// contraction over rows).
// ---------------------------------------------------------------------------------------------------------------------
// stage NKB*32 rows x 64 d of `src` (row stride `stride` elements) as a row image; rows >= N are zero
template <int NKB>
__device__ __forceinline__ void stage_rows(unsigned char* img, const uint16_t* src, long stride, int N, int tid) {
  for (int c = tid; c < NKB * 32 * (kD / 8); c += 256) {
    const int k = c >> 3, d8 = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < N) v = *reinterpret_cast<const uint4*>(src + k * stride + d8 * 8);
    *reinterpret_cast<uint4*>(img + ((((k >> 5) * kKS + (d8 >> 1)) * 64) + (d8 & 1) * 32 + (k & 31)) * 16) = v;
  }
}
// the same rows as a transposed image: img[(blk, t, db)][lane = half*32 + d%32][e] = src[blk*32 + (e&3) + 8*(2t + (e>>2)) + 4*half][db*32 + d%32]
template <int NKB>
__device__ __forceinline__ void stage_transposed(uint16_t* img, const uint16_t* src, long stride, int N, int tid) {
  for (int c = tid; c < NKB * 32 * (kD / 8); c += 256) {
    const int k = c >> 3, d8 = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < N) v = *reinterpret_cast<const uint4*>(src + k * stride + d8 * 8);
    const int blk = k >> 5, kk = k & 31;
    const int vh = (kk >> 2) & 1, t = kk >> 4, e = (kk & 3) + 4 * ((kk >> 3) & 1);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = d8 * 8 + j;
      img[((((blk * 2 + t) * kDB + (d >> 5)) * 64) + vh * 32 + (d & 31)) * 8 + e] =
          static_cast<uint16_t>((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
    }
  }
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// VG (N <= 224): the V row operands of dP = dO V^T come straight from memory (16 contiguous bytes per lane, the (b, h) slice is
// L2-resident), one key block ahead of the MFMAs that use them: 56 instead of 84 KB of LDS, two workgroups per CU.
template <int NKB, bool VG>
__global__ __launch_bounds__(256, (VG ? 2 : 1)) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                             const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             uint16_t* __restrict__ dqkv, float* __restrict__ dvec, int N, int H,
                                                             float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* kr = lds;                                            // K row image
  unsigned char* vr = lds + NKB * kKS * 1024;                         // V row image (not with VG)
  uint16_t* kt = reinterpret_cast<uint16_t*>(lds + (VG ? 1 : 2) * NKB * kKS * 1024);   // K transposed image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long ts = 3L * H * kD, os = static_cast<long>(H) * kD;
  const uint16_t* qb = qkv + static_cast<long>(b) * N * ts + h * kD;
  const uint16_t* kb_ = qb + os;
  const uint16_t* vb = qb + 2 * os;
  const uint16_t* ob = o + static_cast<long>(b) * N * os + h * kD;
  const uint16_t* dob = dout + static_cast<long>(b) * N * os + h * kD;
  stage_rows<NKB>(kr, kb_, ts, N, tid);
  if constexpr (!VG) stage_rows<NKB>(vr, vb, ts, N, tid);
  stage_transposed<NKB>(kt, kb_, ts, N, tid);
  __syncthreads();
  const float c2 = scale * 1.4426950408889634f;
  // VG: lane (l32, half) of key block kblk reads V[min(kblk*32 + l32, N-1)][ks*16 + half*8 .. +8] (rows past N only meet p = 0)
  auto v_global = [&](int kblk, bf16x8* dst) {
    const int kr_ = min(kblk * 32 + l32, N - 1);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(vb + kr_ * ts + ks * 16 + half * 8);
  };
  const unsigned char* krf = kr + lane * 16;
  const unsigned char* vrf = vr + lane * 16;
  const uint16_t* ktf = kt + lane * 8;
  const int nqb = (N + 31) / 32;
  for (int qblk = wave; qblk < nqb; qblk += 4) {
    asm volatile("" : "+v"(krf), "+v"(vrf), "+v"(ktf));
    int q = qblk * 32 + l32;
    const bool q_ok = q < N;
    if (!q_ok) q = N - 1;
    bf16x8 qf[kKS], dof[kKS];
    float dq_part = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qb + q * ts + ks * 16 + half * 8);
      const uint4 dv4 = *reinterpret_cast<const uint4*>(dob + q * os + ks * 16 + half * 8);
      const uint4 ov4 = *reinterpret_cast<const uint4*>(ob + q * os + ks * 16 + half * 8);
      dof[ks] = __builtin_bit_cast(bf16x8, dv4);
      const uint32_t a[4] = {dv4.x, dv4.y, dv4.z, dv4.w}, c[4] = {ov4.x, ov4.y, ov4.z, ov4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) dq_part += bf_lo(a[j]) * bf_lo(c[j]) + bf_hi(a[j]) * bf_hi(c[j]);
    }
    const float Dq = dq_part + __shfl_xor(dq_part, 32, 64);
    const float l2 = lse[(static_cast<long>(b) * H + h) * N + q] * 1.4426950408889634f;
    if (q_ok && half == 0) dvec[(static_cast<long>(b) * H + h) * N + q] = Dq;
    f32x16 dq[kDB];
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    bf16x8 vnext[kKS];
    if constexpr (VG) v_global(0, vnext);
#pragma unroll 1
    for (int kblk = 0; kblk < NKB; ++kblk) {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      bf16x8 vcur[kKS];
      if constexpr (VG) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) vcur[ks] = vnext[ks];
        v_global(min(kblk + 1, NKB - 1), vnext);
      }
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        const bf16x8 ka = *reinterpret_cast<const bf16x8*>(krf + (kblk * kKS + ks) * 1024);
        bf16x8 va;
        if constexpr (VG) va = vcur[ks];
        else va = *reinterpret_cast<const bf16x8*>(vrf + (kblk * kKS + ks) * 1024);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dpt, 0, 0, 0);
      }
      uint32_t pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int key = kblk * 32 + ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * half;
          const float pv = key < N ? __builtin_amdgcn_exp2f(fmaf(st[r + u], c2, -l2)) : 0.f;
          ds[u] = pv * (dpt[r + u] - Dq) * scale;
        }
        pk[r >> 1] = pack_bf16(ds[0], ds[1]);
      }
      const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])),
                             __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7]))};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
          const bf16x8 kb8 = *reinterpret_cast<const bf16x8*>(ktf + ((kblk * 2 + t) * kDB + db) * 512);
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf[t], kb8, dq[db], 0, 0, 0);
        }
    }
    uint16_t* dqb = dqkv + static_cast<long>(b) * N * ts + h * kD;
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (qr < N) dqb[static_cast<long>(qr) * ts + db * 32 + l32] = static_cast<uint16_t>(pack_bf16(dq[db][r], 0.f));
      }
  }
}

// ROWS_LDS = false (N > 224): four operand images of 13 blocks would be 208 KiB, so only the two TRANSPOSED images live in LDS
// and the row operands of S = Q K^T and dP = dO V^T - 16 contiguous bytes per lane - come straight from memory (the (b, h)
// slice is L2-resident: each wavefront re-reads it once per key block), one query block ahead of the MFMAs that use them.
template <int NKB, bool ROWS_LDS>
__global__ __launch_bounds__(256, ((NKB <= 7 && !ROWS_LDS) ? 2 : 1)) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ dvec,
                                                              uint16_t* __restrict__ dqkv, int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int IMG = NKB * kKS * 1024;
  constexpr int NROW = ROWS_LDS ? 2 : 0;                              // row images in LDS
  unsigned char* qr_ = lds;                                           // Q row image
  unsigned char* dor = lds + IMG;                                     // dO row image
  uint16_t* qt = reinterpret_cast<uint16_t*>(lds + NROW * IMG);       // Q transposed
  uint16_t* dot_ = reinterpret_cast<uint16_t*>(lds + (NROW + 1) * IMG);   // dO transposed
  float* lse_s = reinterpret_cast<float*>(lds + (NROW + 2) * IMG);    // [NKB*32] lse * log2(e)
  float* dv_s = lse_s + NKB * 32;                                     // [NKB*32] D_q
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long ts = 3L * H * kD, os = static_cast<long>(H) * kD;
  const uint16_t* qb = qkv + static_cast<long>(b) * N * ts + h * kD;
  const uint16_t* kb_ = qb + os;
  const uint16_t* vb = qb + 2 * os;
  const uint16_t* dob = dout + static_cast<long>(b) * N * os + h * kD;
  if constexpr (ROWS_LDS) {
    stage_rows<NKB>(qr_, qb, ts, N, tid);
    stage_rows<NKB>(dor, dob, os, N, tid);
  }
  stage_transposed<NKB>(qt, qb, ts, N, tid);
  stage_transposed<NKB>(dot_, dob, os, N, tid);
  for (int i = tid; i < NKB * 32; i += 256) {
    lse_s[i] = i < N ? lse[(static_cast<long>(b) * H + h) * N + i] * 1.4426950408889634f : INFINITY;   // exp2(-inf) = 0: padded queries
    dv_s[i] = i < N ? dvec[(static_cast<long>(b) * H + h) * N + i] : 0.f;
  }
  __syncthreads();
  const float c2 = scale * 1.4426950408889634f;
  const unsigned char* qrf = qr_ + lane * 16;
  const unsigned char* dorf = dor + lane * 16;
  const uint16_t* qtf = qt + lane * 8;
  const uint16_t* dotf = dot_ + lane * 8;
  const int nkb = (N + 31) / 32;
  for (int kblk = wave; kblk < nkb; kblk += 4) {
    asm volatile("" : "+v"(qrf), "+v"(dorf), "+v"(qtf), "+v"(dotf));
    int k = kblk * 32 + l32;
    if (k >= N) k = N - 1;                                            // results of padded keys are never stored
    bf16x8 kf[kKS], vf[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kb_ + k * ts + ks * 16 + half * 8);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vb + k * ts + ks * 16 + half * 8);
    }
    f32x16 dk[kDB], dv[kDB];
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    // ROWS_LDS = false: this lane's rows of Q and dO for query block qblk (queries past N are clamped: their lse is +inf,
    // so P and dS vanish whatever the operands)
    bf16x8 qn[kKS], dn[kKS];
    auto load_rows = [&](int qblk) {
      int q = qblk * 32 + l32;
      if (q >= N) q = N - 1;
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        qn[ks] = *reinterpret_cast<const bf16x8*>(qb + q * ts + ks * 16 + half * 8);
        dn[ks] = *reinterpret_cast<const bf16x8*>(dob + static_cast<long>(q) * os + ks * 16 + half * 8);
      }
    };
    if constexpr (!ROWS_LDS) load_rows(0);
#pragma unroll 1
    for (int qblk = 0; qblk < (ROWS_LDS ? NKB : nkb); ++qblk) {
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
      bf16x8 qc[kKS], dc[kKS];
      if constexpr (!ROWS_LDS) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) { qc[ks] = qn[ks]; dc[ks] = dn[ks]; }
        if (qblk + 1 < nkb) load_rows(qblk + 1);                       // in flight behind this block's 24 MFMAs
      }
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        const bf16x8 qa = ROWS_LDS ? *reinterpret_cast<const bf16x8*>(qrf + (qblk * kKS + ks) * 1024) : qc[ks];
        const bf16x8 da = ROWS_LDS ? *reinterpret_cast<const bf16x8*>(dorf + (qblk * kKS + ks) * 1024) : dc[ks];
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sc, 0, 0, 0);      // D[i = q][j = k]: lane = key
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
      }
      uint32_t pp[8], pd[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float pv[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qi = qblk * 32 + ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * half;
          pv[u] = __builtin_amdgcn_exp2f(fmaf(sc[r + u], c2, -lse_s[qi]));
          ds[u] = pv[u] * (dp[r + u] - dv_s[qi]) * scale;
        }
        pp[r >> 1] = pack_bf16(pv[0], pv[1]);
        pd[r >> 1] = pack_bf16(ds[0], ds[1]);
      }
      const bf16x8 pf[2] = {__builtin_bit_cast(bf16x8, make_uint4(pp[0], pp[1], pp[2], pp[3])),
                            __builtin_bit_cast(bf16x8, make_uint4(pp[4], pp[5], pp[6], pp[7]))};
      const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, make_uint4(pd[0], pd[1], pd[2], pd[3])),
                             __builtin_bit_cast(bf16x8, make_uint4(pd[4], pd[5], pd[6], pd[7]))};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int db = 0; db < kDB; ++db) {
          const bf16x8 dob8 = *reinterpret_cast<const bf16x8*>(dotf + ((qblk * 2 + t) * kDB + db) * 512);
          const bf16x8 qb8 = *reinterpret_cast<const bf16x8*>(qtf + ((qblk * 2 + t) * kDB + db) * 512);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[t], dob8, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf[t], qb8, dk[db], 0, 0, 0);
        }
    }
    uint16_t* dkb = dqkv + static_cast<long>(b) * N * ts + os + h * kD;
    uint16_t* dvb = dkb + os;
#pragma unroll
    for (int db = 0; db < kDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr2 = kblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (kr2 < N) {
          dkb[static_cast<long>(kr2) * ts + db * 32 + l32] = static_cast<uint16_t>(pack_bf16(dk[db][r], 0.f));
          dvb[static_cast<long>(kr2) * ts + db * 32 + l32] = static_cast<uint16_t>(pack_bf16(dv[db][r], 0.f));
        }
      }
  }
}

template <int NKB>
int launch_attn_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int N, int H, float scale, hipStream_t s) {
  const size_t lds = static_cast<size_t>(NKB) * (kKS + 2 * kDB) * 1024;
  auto kfn = attn_fwd_kernel<NKB>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(static_cast<unsigned>(B * H)), dim3(256), lds, s, qkv, out, lse, N, H,
                     scale * 1.4426950408889634f);
  return launch_status();
}

template <int NKB, bool ROWS_LDS>
int launch_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, uint16_t* dqkv, float* dvec,
                    int64_t B, int32_t N, int32_t H, float scale, hipStream_t s) {
  constexpr bool vg_on = true;
  constexpr bool VG = NKB <= 7;
  const bool vg = VG && vg_on;
  const size_t lds_q = static_cast<size_t>(NKB) * ((vg ? 1 : 2) * kKS + 2 * kDB) * 1024;
  const size_t lds_kv = static_cast<size_t>(NKB) * (ROWS_LDS ? 4 : 2) * kKS * 1024 + 2 * NKB * 32 * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<NKB, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(static_cast<size_t>(NKB) * (2 * kKS + 2 * kDB) * 1024));
    if constexpr (VG)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<NKB, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(static_cast<size_t>(NKB) * (kKS + 2 * kDB) * 1024));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<NKB, ROWS_LDS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_kv));
    attr_done = true;
  }
  const dim3 grid(static_cast<unsigned>(B * H)), block(256);
  if constexpr (VG) {
    if (vg) hipLaunchKernelGGL((attn_bwd_dq_kernel<NKB, true>), grid, block, lds_q, s, qkv, out, dout, lse, dqkv, dvec, N, H, scale);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<NKB, false>), grid, block, lds_q, s, qkv, out, dout, lse, dqkv, dvec, N, H, scale);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NKB, false>), grid, block, lds_q, s, qkv, out, dout, lse, dqkv, dvec, N, H, scale);
  }
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKB, ROWS_LDS>), grid, block, lds_kv, s, qkv, dout, lse, dvec, dqkv, N, H, scale);
  return launch_status();
}

}  // namespace

extern "C" {

int cnx_attention_supported(int32_t N, int32_t head_dim) { return (head_dim == kD && N >= 1 && N <= 416) ? 1 : 0; }

int cnx_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int32_t N, int32_t H, int32_t head_dim, float scale,
                      void* stream) {
  if (B < 0 || N <= 0 || H <= 0) return APGD_ERR_SIZE;
  if (B == 0) return APGD_OK;
  if (!qkv || !out) return APGD_ERR_NULL;
  if (!cnx_attention_supported(N, head_dim)) return APGD_ERR_ARG;
  if (B * H > 0x7fffffff) return APGD_ERR_SIZE;
  hipStream_t s = as_stream(stream);
  const auto* q = static_cast<const uint16_t*>(qkv);
  auto* o = static_cast<uint16_t*>(out);
  if (N <= 224) return launch_attn_fwd<7>(q, o, lse, B, N, H, scale, s);
  return launch_attn_fwd<13>(q, o, lse, B, N, H, scale, s);
}

int cnx_attention_bwd_supported(int32_t N, int32_t head_dim) { return (head_dim == kD && N >= 1 && N <= 416) ? 1 : 0; }


int cnx_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dvec,
                      int64_t B, int32_t N, int32_t H, int32_t head_dim, float scale, void* stream) {
  if (B < 0 || N <= 0 || H <= 0) return APGD_ERR_SIZE;
  if (B == 0) return APGD_OK;
  if (!qkv || !out || !dout || !lse || !dqkv || !dvec) return APGD_ERR_NULL;
  if (!cnx_attention_bwd_supported(N, head_dim)) return APGD_ERR_ARG;
  if (B * H > 0x7fffffff) return APGD_ERR_SIZE;
  hipStream_t s = as_stream(stream);
  const auto* q = static_cast<const uint16_t*>(qkv);
  const auto* o = static_cast<const uint16_t*>(out);
  const auto* d = static_cast<const uint16_t*>(dout);
  auto* g = static_cast<uint16_t*>(dqkv);
  // N <= 224: row operands from memory too (56 KB of LDS and 246 registers: two workgroups per CU; with the row images in LDS -
  // 112 KB, 303 registers - it is one); APGD_ATTN_DKV_ROWS=1 selects the LDS-row form
  constexpr bool rows_lds = false;
  if (N <= 224) return rows_lds ? launch_attn_bwd<7, true>(q, o, d, lse, g, dvec, B, N, H, scale, s)
                                : launch_attn_bwd<7, false>(q, o, d, lse, g, dvec, B, N, H, scale, s);
  return launch_attn_bwd<13, false>(q, o, d, lse, g, dvec, B, N, H, scale, s);
}

}  // extern "C"
