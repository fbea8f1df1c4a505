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

}  // extern "C"
