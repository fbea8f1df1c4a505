// stem_kernels.hip — first ConvStem convolution (3 -> P channels, 3x3, stride 2, pad 1; utils_architecture.py:127, 180, 205)
// for gfx950, forward and input gradient.
//
// This layer touches the attack state directly: its input is the fp32 NCHW image batch x_adv and its input gradient IS
// the gradient the APGD update consumes.  Through the library it costs a cast + a layout copy + an implicit-GEMM kernel
// + a separate bias add in the forward (380 us at batch 256) and a layout copy + CK conv_bwd_data + two more copies in
// the backward (710 us), for 4.2 GFLOP of work.  Here:
//   forward   a wavefront = 32 output positions: the 27 taps are read straight from the fp32 NCHW image (rounded to bf16 as
//             the autocast convolution does) into an MFMA A operand, the filter sits in registers as B fragments, the
//             result (+bias) goes through a wavefront-private LDS transpose to lane-linear 16-byte NHWC stores;
//   dgrad     a wavefront = 32 input 2x2 patches (all 3 channels): with stride 2 the four pixels of a patch see the 2x2
//             neighbouring output positions through fixed (kh, kw) taps - a 12 x 4P matrix times the 4P gradient values
//             (16-byte NHWC loads), transposed product so that a lane owns a patch and stores float2 runs of fp32 NCHW.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float round_bf16(float v) { return __uint_as_float(pack_bf16(v, 0.f) << 16); }
// exact-erf GELU, the evaluation shared by all kernels of the library (tools/fit_gelu.py: |error| <= 1.2e-6)
__device__ __forceinline__ float gelu_f(float z) {
  const float az = fabsf(z);
  float q = fmaf(-0.00041175442346105595f, az, 0.006678475199902348f);
  q = fmaf(q, az, -0.050879760394516485f);
  q = fmaf(q, az, -0.46094072908550926f);
  q = fmaf(q, az, -1.150400682855232f);
  q = fmaf(q, az, -8.454223479528131e-05f);
  return fmaf(az * __builtin_amdgcn_exp2f(q), -0.5f, fmaxf(z, 0.0f));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Both directions are tiny GEMMs per group of 32 positions, on MFMA 32x32x16 (the fp32-FMA form of these kernels was
// VALU-bound at 420 / 780 us; 1296 MACs per position against 27 loads):
//   forward  D[pos][co]  = patch[pos][k] W[k][co],  k = 4 (ci * 3 + kh) + kw, kw = 3 a zero column: 36 padded to 48 = 3 k-steps x
//            ceil(P/32) blocks.  The padding buys 16-byte loads: the three taps of a (ci, kh) are consecutive fp32 of one image row,
//            a lane fetches them (and the float behind them, which meets a zero weight) with ONE instruction - 6 loads per lane and
//            32 positions instead of 16 scalar ones with their per-element address arithmetic (the first form of this kernel:
//            166 us; the instruction count of its loads and their integer divisions were the bound, not the 462 MB it moves)
//   dgrad    D[o][patch] = Wp[o][k] dY[k][patch],   k = (q, co): the 2x2 output positions q around a 2x2 input patch times the
//            P channels; o = ci*4 + i*2 + j: the 12 values of the patch.  Wp[o][(q, co)] = w[co][ci][kh][kw] with the (kh, kw)
//            that connects pixel (i, j) to position q (stride 2: at most one), else 0.                  4P/16 k-steps
// Packed filter (cnx_stem_conv_pack): [NB][3][64][8] bf16 forward B fragments, then [4P/16][64][8] bf16 dgrad A fragments.
template <int P> struct StemGeo {
  static constexpr int NB = (P + 31) / 32;            // 32-wide blocks of output channels
  static constexpr int KD = 4 * P / 16;               // k-steps of the dgrad GEMM
  static constexpr int KF = 3;                        // k-steps of the forward GEMM: 9 (ci, kh) quads of 4 (kw 0..2 + a zero) = 36 -> 48
  static constexpr int FWD_BYTES = NB * KF * 1024;
  static constexpr int BYTES = FWD_BYTES + KD * 1024;
};

template <typename TW>
__global__ void stem_pack_kernel(const TW* __restrict__ w, uint16_t* __restrict__ wq, int P) {
  const int NB = (P + 31) / 32, KD = 4 * P / 16;
  const int i = blockIdx.x * 256 + threadIdx.x;                      // one bf16 element each
  const int n_fwd = NB * 3 * 512, n_all = n_fwd + KD * 512;
  if (i >= n_all) return;
  float v = 0.f;
  if (i < n_fwd) {
    const int e = i & 7, lane = (i >> 3) & 63, piece = i >> 9, ks = piece % 3, nb = piece / 3;
    const int co = nb * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + e;
    const int quad = k >> 2, kw = k & 3;                              // quad = ci * 3 + kh
    if (co < P && quad < 9 && kw < 3) v = static_cast<float>(w[co * 27 + quad * 3 + kw]);
  } else {
    const int t = i - n_fwd;
    const int e = t & 7, lane = (t >> 3) & 63, ks = t >> 9;
    const int o = lane & 31, k = ks * 16 + (lane >> 5) * 8 + e;
    const int q = k / P, co = k - q * P, qi = q >> 1, qj = q & 1;
    if (o < 12) {
      const int ci = o >> 2, pi = (o >> 1) & 1, pj = o & 1;
      // pixel row 2a+pi <- output row a+qi: pi=0: only qi=0 through kh=1;  pi=1: qi=0 through kh=2, qi=1 through kh=0
      const int kh = pi == 0 ? (qi == 0 ? 1 : -1) : (qi == 0 ? 2 : 0);
      const int kw = pj == 0 ? (qj == 0 ? 1 : -1) : (qj == 0 ? 2 : 0);
      if (kh >= 0 && kw >= 0) v = static_cast<float>(w[co * 27 + (ci * 3 + kh) * 3 + kw]);
    }
  }
  wq[i] = static_cast<uint16_t>(pack_bf16(v, 0.f));
}

template <int P, bool LNG>
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wq,
                                                            const float* __restrict__ bias, uint16_t* __restrict__ out, long total,
                                                            int H, int W, int OH, int OW, const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b, float eps, uint16_t* __restrict__ act,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
  using G = StemGeo<P>;
  __shared__ __attribute__((aligned(16))) uint16_t stage[4][32 * P];
  __shared__ __attribute__((aligned(16))) uint16_t stage2[LNG ? 4 : 1][LNG ? 32 * P : 8];
  // LNG: the ConvStem's LayerNorm(channels) + GELU on the 32 x P tile while it sits in LDS (two lanes per position, P/2 channels
  // each): the activation leaves next to (or, for a gradient-free forward, instead of) the convolution output
  constexpr int NCH = P / 16;                                       // 8-channel chunks per lane in the LN phase
  float lw[LNG ? NCH * 8 : 1], lb[LNG ? NCH * 8 : 1];
  if constexpr (LNG) {
    const int h2 = threadIdx.x & 1;
#pragma unroll
    for (int i = 0; i < NCH * 8; ++i) { lw[i] = ln_w[h2 * (P / 2) + i]; lb[i] = ln_b[h2 * (P / 2) + i]; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
  bf16x8 bw[G::NB][G::KF];
#pragma unroll
  for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < G::KF; ++ks) bw[nb][ks] = *reinterpret_cast<const bf16x8*>(wq + ((nb * G::KF + ks) * 64 + lane) * 8);
  // the lane's quads: k-step s, slot t -> quad q = 4 s + 2 half + t (q >= 9: none); element offset of its image row relative to
  // row 2 oh - 1 of channel 0, and whether it is the filter's first row (kh = 0: outside the image for oh = 0)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0,
      static_cast<uint32_t>((total / (static_cast<long>(OH) * OW)) * 3 * H * W * 4), 0x00020000);
  int qoff[G::KF][2];
  bool qtop[G::KF][2], qhas[G::KF][2];
#pragma unroll
  for (int sk = 0; sk < G::KF; ++sk)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int q = 4 * sk + 2 * half + t;
      const int ci = q / 3, kh = q - 3 * ci;
      qoff[sk][t] = (ci * H + kh) * W;
      qtop[sk][t] = kh == 0;
      qhas[sk][t] = q < 9;
    }
  float bv[G::NB];
#pragma unroll
  for (int nb = 0; nb < G::NB; ++nb) bv[nb] = (bias && nb * 32 + l32 < P) ? bias[nb * 32 + l32] : 0.f;
  const long nblk = (total + 31) / 32;
  for (long blk = static_cast<long>(blockIdx.x) * 4 + wave; blk < nblk; blk += static_cast<long>(gridDim.x) * 4) {
    long pos = blk * 32 + l32;
    const bool ok = pos < total;
    if (!ok) pos = total - 1;
    const int ow = static_cast<int>(pos % OW);
    const long r = pos / OW;
    const int oh = static_cast<int>(r % OH);
    // element index of (channel 0, row 2 oh - 1, column 2 ow - 1) of this image: may be negative (first row / column); a quad outside
    // the image or absent is fetched from offset 0xffffffff - beyond the buffer's range, the hardware returns zeros
    const long e0 = (r / OH) * 3 * static_cast<long>(H) * W + static_cast<long>(2 * oh - 1) * W + (2 * ow - 1);
    bf16x8 pa[G::KF];
#pragma unroll
    for (int sk = 0; sk < G::KF; ++sk) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4s;
      u32x4s qv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool in = qhas[sk][t] && !(qtop[sk][t] && oh == 0);
        // (ow = 0: the quad would start at column -1 - for the tensor's very first row at element -1; it is fetched from column 0
        //  and shifted instead)
        const uint32_t off = in ? static_cast<uint32_t>(e0 + qoff[sk][t] + (ow == 0 ? 1 : 0)) * 4u : 0xffffffffu;
        qv[t] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
        if (ow == 0) { qv[t].z = qv[t].y; qv[t].y = qv[t].x; qv[t].x = 0u; }
        qv[t].w = 0u;                                                  // the quad's 4th value is a NEIGHBOUR (next column / row / image) under
                                                                       // a zero weight: 0 * Inf would be NaN in this position's output
      }
      pa[sk] = __builtin_bit_cast(bf16x8, make_uint4(pack_bf16(__uint_as_float(qv[0].x), __uint_as_float(qv[0].y)),
                                                     pack_bf16(__uint_as_float(qv[0].z), __uint_as_float(qv[0].w)),
                                                     pack_bf16(__uint_as_float(qv[1].x), __uint_as_float(qv[1].y)),
                                                     pack_bf16(__uint_as_float(qv[1].z), __uint_as_float(qv[1].w))));
    }
    uint16_t* st = stage[wave];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb) {
      f32x16 acc;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[rr] = bv[nb];
#pragma unroll
      for (int sk = 0; sk < G::KF; ++sk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[sk], bw[nb][sk], acc, 0, 0, 0);
      const int co = nb * 32 + l32;
      if (co < P) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) st[((rr & 3) + 8 * (rr >> 2) + 4 * half) * P + co] = static_cast<uint16_t>(pack_bf16(acc[rr], 0.f));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the region is private to the wavefront
    uint16_t* st2 = stage2[LNG ? wave : 0];
    if constexpr (LNG) {
      const int pl = lane >> 1, h2 = lane & 1;
      const uint16_t* src = st + pl * P + h2 * (P / 2);
      float v[NCH * 8];
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const uint4 t = *reinterpret_cast<const uint4*>(src + k * 8);
        const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k * 8 + 2 * e] = __uint_as_float(w4[e] << 16); v[k * 8 + 2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
      }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NCH * 8; ++i) sm += v[i];
      sm += __shfl_xor(sm, 1, 64);
      const float mu = sm * (1.0f / P);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH * 8; ++i) { const float d = v[i] - mu; q += d * d; }
      q += __shfl_xor(q, 1, 64);
      const float rs = rsqrtf(q * (1.0f / P) + eps);
      const long posl = blk * 32 + pl;
      if (h2 == 0 && mean && posl < total) { mean[posl] = mu; rstd[posl] = rs; }
      uint16_t* dst = st2 + pl * P + h2 * (P / 2);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gelu_f((v[k * 8 + e] - mu) * rs * lw[k * 8 + e] + lb[k * 8 + e]);
        *reinterpret_cast<uint4*>(dst + k * 8) = make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    // 32 positions x P channels = 64 P contiguous bytes in NHWC memory: P/8 16-byte chunks per position, lane-linear
    const long base = blk * 32 * P;                                  // element offset of the block
    const long lim = total * P;
#pragma unroll
    for (int c = 0; c < (32 * P / 8 + 63) / 64; ++c) {
      const int ch = c * 64 + lane;
      if (ch < 32 * P / 8 && base + static_cast<long>(ch) * 8 < lim) {
        if (out) *reinterpret_cast<uint4*>(out + base + static_cast<long>(ch) * 8) = *reinterpret_cast<const uint4*>(st + ch * 8);
        if constexpr (LNG) *reinterpret_cast<uint4*>(act + base + static_cast<long>(ch) * 8) = *reinterpret_cast<const uint4*>(st2 + ch * 8);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

// TO = float: the input gradient itself;  TO = int8_t: its SIGN {-1, 0, +1} (sign(NaN) = 0, as torch.sign on the
// reference's path) - all the Linf update reads of the gradient (autopgd_train_clean.py:221), a quarter of the bytes.
// BLK (int8 signs only): the "blocked" order the Linf update kernel reads 16 signs per lane in (apgd_hip.h, APGD_I8_BLK) - inside
// every group of 1024 elements of a sample, bits [2..9] of the element index, (u : 2 bits, lane : 6 bits), are stored as
// (lane, u): element g*1024 + (u*64 + lane)*4 + j lives at byte g*1024 + lane*16 + u*4 + j.  Pairs (j even, j + 1) stay adjacent.
__device__ __forceinline__ long blk_index(long e) {
  return (e & ~1023L) | (((e >> 2) & 63) << 4) | (((e >> 8) & 3) << 2) | (e & 3);
}
template <typename TO, bool BLK>
__device__ __forceinline__ void st_pair(TO* base, long e, float a, float b) {
  if constexpr (sizeof(TO) == 4) {
    *reinterpret_cast<float2*>(base + e) = make_float2(a, b);
  } else {
    const int sa = (a > 0.f) - (a < 0.f), sb = (b > 0.f) - (b < 0.f);
    *reinterpret_cast<uint16_t*>(base + (BLK ? blk_index(e) : e)) = static_cast<uint16_t>((sa & 0xff) | ((sb & 0xff) << 8));
  }
}

template <int P, typename TO, bool BLK = false>
__global__ __launch_bounds__(256) void stem_conv_dgrad_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ wq,
                                                              TO* __restrict__ dx, long total, int H, int W, int OH, int OW) {
  using G = StemGeo<P>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
  bf16x8 wa[G::KD];
#pragma unroll
  for (int ks = 0; ks < G::KD; ++ks) wa[ks] = *reinterpret_cast<const bf16x8*>(wq + G::FWD_BYTES / 2 + (ks * 64 + lane) * 8);
  const long nblk = (total + 31) / 32;
  for (long blk = static_cast<long>(blockIdx.x) * 4 + wave; blk < nblk; blk += static_cast<long>(gridDim.x) * 4) {
    long pt = blk * 32 + l32;
    const bool ok = pt < total;
    if (!ok) pt = total - 1;
    const int b = static_cast<int>(pt % OW);
    const long r = pt / OW;
    const int a = static_cast<int>(r % OH);
    const long n = r / OH;
    const uint16_t* p00 = dy + ((n * OH + a) * static_cast<long>(OW) + b) * P;
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KD; ++ks) {
      const int k = ks * 16 + half * 8;                               // 8 consecutive channels of ONE position q (P % 8 == 0)
      const int q = k / P, co = k - q * P, qi = q >> 1, qj = q & 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a + qi < OH && b + qj < OW) v = *reinterpret_cast<const uint4*>(p00 + (static_cast<long>(qi) * OW + qj) * P + co);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
    }
    // acc[rr] = D[o = (rr&3) + 8*(rr>>2) + 4*half][patch = l32], o = ci*4 + i*2 + j
    if (ok) {
      TO* xn = dx + n * 3 * static_cast<long>(H) * W;
      if (half == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          st_pair<TO, BLK>(xn, (0L * H + 2 * a + i) * W + 2 * b, acc[2 * i], acc[2 * i + 1]);            // ci = 0: o = 0..3
          st_pair<TO, BLK>(xn, (2L * H + 2 * a + i) * W + 2 * b, acc[4 + 2 * i], acc[4 + 2 * i + 1]);    // ci = 2: o = 8..11
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          st_pair<TO, BLK>(xn, (1L * H + 2 * a + i) * W + 2 * b, acc[2 * i], acc[2 * i + 1]);            // ci = 1: o = 4..7
      }
    }
  }
}

}  // namespace

extern "C" {

int cnx_stem_conv_supported(int32_t P) { return (P == 48 || P == 64 || P == 96) ? 1 : 0; }

int64_t cnx_stem_conv_packed_bytes(int32_t P) { return (static_cast<int64_t>((P + 31) / 32) * 3 + 4 * P / 16) * 1024; }   // StemGeo<P>::BYTES

int cnx_stem_conv_pack(const void* w, int w_dtype, void* wq, int32_t P, void* stream) {
  if (!w || !wq) return APGD_ERR_NULL;
  if (!cnx_stem_conv_supported(P)) return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const long n = cnx_stem_conv_packed_bytes(P) / 2;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
  auto* q = static_cast<uint16_t*>(wq);
  if (w_dtype == APGD_F32) hipLaunchKernelGGL(stem_pack_kernel<float>, grid, block, 0, s, static_cast<const float*>(w), q, P);
  else if (w_dtype == APGD_BF16) hipLaunchKernelGGL(stem_pack_kernel<__bf16>, grid, block, 0, s, static_cast<const __bf16*>(w), q, P);
  else return APGD_ERR_DTYPE;
  return launch_status();
}

int cnx_stem_conv_fwd(const float* x, const void* wq, const float* bias, void* out, int64_t N, int32_t H, int32_t W, int32_t P,
                      void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!x || !wq || !out) return APGD_ERR_NULL;
  if (!cnx_stem_conv_supported(P)) return APGD_ERR_ARG;
  if ((H & 1) || (W & 1) || static_cast<long>(N) * 3 * H * W >= (1L << 30)) return APGD_ERR_ARG;   // even maps; 32-bit byte offsets (buffer loads)
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long total = N * OH * OW;
  long nb = (total + 127) / 128;
  if (nb > 4096) nb = 4096;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t s = as_stream(stream);
  auto* o = static_cast<uint16_t*>(out);
  const auto* q = static_cast<const uint16_t*>(wq);
  const float* nf = nullptr;
  uint16_t* nu = nullptr;
  float* nm = nullptr;
  if (P == 48) hipLaunchKernelGGL((stem_conv_fwd_kernel<48, false>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, nf, nf, 0.f, nu, nm, nm);
  else if (P == 64) hipLaunchKernelGGL((stem_conv_fwd_kernel<64, false>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, nf, nf, 0.f, nu, nm, nm);
  else hipLaunchKernelGGL((stem_conv_fwd_kernel<96, false>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, nf, nf, 0.f, nu, nm, nm);
  return launch_status();
}

int cnx_stem_conv_ln_gelu_fwd(const float* x, const void* wq, const float* bias, const float* ln_w, const float* ln_b, float eps,
                              void* y, void* act, float* mean, float* rstd, int64_t N, int32_t H, int32_t W, int32_t P,
                              void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!x || !wq || !ln_w || !ln_b || !act) return APGD_ERR_NULL;
  if ((mean == nullptr) != (rstd == nullptr)) return APGD_ERR_ARG;
  if (!cnx_stem_conv_supported(P)) return APGD_ERR_ARG;
  if ((H & 1) || (W & 1) || static_cast<long>(N) * 3 * H * W >= (1L << 30)) return APGD_ERR_ARG;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long total = N * OH * OW;
  long nb = (total + 127) / 128;
  if (nb > 4096) nb = 4096;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t s = as_stream(stream);
  auto* o = static_cast<uint16_t*>(y);
  auto* a = static_cast<uint16_t*>(act);
  const auto* q = static_cast<const uint16_t*>(wq);
  if (P == 48) hipLaunchKernelGGL((stem_conv_fwd_kernel<48, true>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, ln_w, ln_b, eps, a, mean, rstd);
  else if (P == 64) hipLaunchKernelGGL((stem_conv_fwd_kernel<64, true>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, ln_w, ln_b, eps, a, mean, rstd);
  else hipLaunchKernelGGL((stem_conv_fwd_kernel<96, true>), grid, block, 0, s, x, q, bias, o, total, H, W, OH, OW, ln_w, ln_b, eps, a, mean, rstd);
  return launch_status();
}

static int stem_dgrad_impl(const void* dy, const void* wq, void* dx, int sign, int64_t N, int32_t H, int32_t W, int32_t P, void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!dy || !wq || !dx) return APGD_ERR_NULL;
  if (!cnx_stem_conv_supported(P) || (H & 1) || (W & 1)) return APGD_ERR_ARG;
  const int OH = H / 2, OW = W / 2;
  const long total = N * OH * OW;
  long nb = (total + 127) / 128;
  if (nb > 4096) nb = 4096;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t s = as_stream(stream);
  const auto* d = static_cast<const uint16_t*>(dy);
  const auto* q = static_cast<const uint16_t*>(wq);
  if (sign == 2) {                                   // blocked sign order (groups of 1024 elements per sample)
    if ((3L * H * W) % 1024 != 0) return APGD_ERR_ARG;
    auto* o = static_cast<int8_t*>(dx);
    if (P == 48) hipLaunchKernelGGL((stem_conv_dgrad_kernel<48, int8_t, true>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else if (P == 64) hipLaunchKernelGGL((stem_conv_dgrad_kernel<64, int8_t, true>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else hipLaunchKernelGGL((stem_conv_dgrad_kernel<96, int8_t, true>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
  } else if (sign) {
    auto* o = static_cast<int8_t*>(dx);
    if (P == 48) hipLaunchKernelGGL((stem_conv_dgrad_kernel<48, int8_t>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else if (P == 64) hipLaunchKernelGGL((stem_conv_dgrad_kernel<64, int8_t>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else hipLaunchKernelGGL((stem_conv_dgrad_kernel<96, int8_t>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
  } else {
    auto* o = static_cast<float*>(dx);
    if (P == 48) hipLaunchKernelGGL((stem_conv_dgrad_kernel<48, float>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else if (P == 64) hipLaunchKernelGGL((stem_conv_dgrad_kernel<64, float>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
    else hipLaunchKernelGGL((stem_conv_dgrad_kernel<96, float>), grid, block, 0, s, d, q, o, total, H, W, OH, OW);
  }
  return launch_status();
}

int cnx_stem_conv_dgrad(const void* dy, const void* wq, float* dx, int64_t N, int32_t H, int32_t W, int32_t P, void* stream) {
  return stem_dgrad_impl(dy, wq, dx, 0, N, H, W, P, stream);
}

int cnx_stem_conv_dgrad_sign(const void* dy, const void* wq, int8_t* sign_out, int64_t N, int32_t H, int32_t W, int32_t P,
                             void* stream) {
  return stem_dgrad_impl(dy, wq, sign_out, 1, N, H, W, P, stream);
}

int cnx_stem_conv_dgrad_sign_blk(const void* dy, const void* wq, int8_t* sign_out, int64_t N, int32_t H, int32_t W, int32_t P,
                                 void* stream) {
  return stem_dgrad_impl(dy, wq, sign_out, 2, N, H, W, P, stream);
}

}  // extern "C"
