// conv2_kernels.hip - the second ConvStem convolution (utils_architecture.py:207-209 ConvBlock1: 48 -> 96; ConvBlock3's 64 -> 96:
// 3x3, stride 2, padding 1) on channels-last bf16 activations for gfx950: forward and input gradient as implicit GEMMs on MFMA.
//
// Through the library this layer is a CK grouped-conv forward (186 us at batch 256, 112x112x48 -> 56x56x96) and an implicit-GEMM
// backward-data kernel (304 us) whose result is not reproducible from run to run (profiles/r02_determinism.log); both move
// 462 MB, i.e. ~95 us at what HBM delivers.
//
//   forward   a wavefront = a 4 x 8 block of output positions.  D[pos][co] = sum_k A[pos][k] B[k][co], k = (tap, ci):
//             A fragments are 16-byte loads straight from the NHWC input (a lane = one position, 8 consecutive channels of one tap;
//             zero outside the image), B fragments = the whole filter, resident in LDS in fragment order (9 * CI/16 * CO/32 KiB).
//             In the accumulator a lane owns one output channel, so bias add and the NHWC store are 64 contiguous bytes per
//             half-wave and position.
//   dgrad     a wavefront = a 4 x 8 block of input positions of ONE parity class (iy & 1, ix & 1): with stride 2 a position is
//             reached by 1, 2, 2 or 4 taps depending on its parity; D[pos][ci] = sum_k dY[pos'][k] Wt[k][ci], k = (tap, co).
//             A fragments from the NHWC output gradient, B fragments (transposed filter, per tap) resident in LDS.
// One workgroup of 16 wavefronts per CU shares the filter; wavefronts take tiles in a grid-stride loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ uint16_t to_bf16(float v) {
  const f32x2 p = {v, 0.f};
  return static_cast<uint16_t>(__builtin_bit_cast(uint32_t, __builtin_convertvector(p, bf16x2)) & 0xffffu);
}

constexpr int kWaves = 16;

// ------------------------------------------------------------------ filter pre-arrangement
// forward  piece (ks, nb), ks = tap * CI/16 + cg: lane (n = l32, half), e -> W[nb*32 + n][cg*16 + half*8 + e][tap/3][tap%3]
// dgrad    piece (tap, ks, nb), ks over CO/16, nb over ceil(CI/32): lane (n, half), e -> W[ks*16 + half*8 + e][nb*32 + n][tap] (0 if
//          nb*32 + n >= CI)
template <typename TW>
__global__ __launch_bounds__(256) void conv2_pack_kernel(const TW* __restrict__ w, uint16_t* __restrict__ wf, uint16_t* __restrict__ wd,
                                                         int CI, int CO) {
  const int KSF = 9 * (CI / 16), NBF = CO / 32, KSD = CO / 16, NBD = (CI + 31) / 32;
  const long nf = static_cast<long>(KSF) * NBF * 512, nd = 9L * KSD * NBD * 512;
  const long q = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (q < nf) {
    const int e = q & 7, lane = (q >> 3) & 63, piece = static_cast<int>(q >> 9);
    const int nb = piece % NBF, ks = piece / NBF, tap = ks / (CI / 16), cg = ks % (CI / 16);
    const int co = nb * 32 + (lane & 31), ci = cg * 16 + (lane >> 5) * 8 + e;
    wf[q] = to_bf16(static_cast<float>(w[(static_cast<long>(co) * CI + ci) * 9 + tap]));
  } else if (q >= nf + nd && q < nf + nd + 64) {
    wd[q - nf] = 0;                                       // the zero page (read through the same pointer arithmetic as wd)
  } else if (q < nf + nd) {
    const long p = q - nf;
    const int e = p & 7, lane = (p >> 3) & 63, piece = static_cast<int>(p >> 9);
    const int nb = piece % NBD, ks = (piece / NBD) % KSD, tap = piece / (NBD * KSD);
    const int ci = nb * 32 + (lane & 31), co = ks * 16 + (lane >> 5) * 8 + e;
    wd[p] = ci < CI ? to_bf16(static_cast<float>(w[(static_cast<long>(co) * CI + ci) * 9 + tap])) : static_cast<uint16_t>(0);
  }
}

// ------------------------------------------------------------------ forward
// Per wavefront a stream of "row steps" (tile, dy): the 3 * CI/16 operand fragments of filter row dy are loaded one step ahead of the
// MFMAs that use them (two register sets), so the HBM / L2 latency of a step hides behind the previous step's 9 * CI/16 * CO/32
// MFMAs.  D[co][pos] (filter fragment first): a lane owns one position and, per accumulator register group, 4 consecutive channels
// - 8-byte writes into a wavefront-private LDS tile, which leaves as 16 bytes per lane of contiguous NHWC rows.
// (wavefronts per workgroup: 12 where filter + output tiles fit the 160 KiB of LDS - CI = 48: 81 + 72 KiB -, else 8)
template <int CI> struct FwdWaves { static constexpr int value = (9 * (CI / 16) * 3 + 12 * 6 + 1 <= 160) ? 12 : 8; };

template <int CI, int CO, int kFwdWaves = FwdWaves<CI>::value>
__global__ __launch_bounds__(kFwdWaves * 64) void conv2_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wf,
                                                                   const float* __restrict__ bias, uint16_t* __restrict__ out, int N, int H,
                                                                   int W) {
  constexpr int CG = CI / 16, KS = 9 * CG, NB = CO / 32, RF = 3 * CG;          // RF fragments per filter row
  constexpr int WBYTES = KS * NB * 1024, TILE_BYTES = 32 * CO * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* bias_s = reinterpret_cast<float*>(lds + WBYTES);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* scr = lds + WBYTES + CO * 4 + wave * TILE_BYTES;              // this wavefront's output tile [32 pos][CO] bf16
  for (int i = tid; i < KS * NB * 64; i += kFwdWaves * 64) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(wf)[i];
  for (int i = tid; i < CO; i += kFwdWaves * 64) bias_s[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const unsigned char* wl = lds + lane * 16;
  const uint16_t* zero_page = wf + KS * NB * 512 + 9 * (CO / 16) * ((CI + 31) / 32) * 512;     // 64 zero elements behind the packed filter
  const int OH = H / 2, OW = W / 2, TX = OW / 8, TY = OH / 4;
  const long n_tiles = static_cast<long>(N) * TY * TX;
  const long t_first = static_cast<long>(blockIdx.x) * kFwdWaves + wave, t_stride = static_cast<long>(gridDim.x) * kFwdWaves;
  const long my_tiles = t_first < n_tiles ? (n_tiles - t_first + t_stride - 1) / t_stride : 0;
  const long n_steps = 3 * my_tiles;

  // operand fragments of row step `st` -> DST
#define C2_LOAD(DST, ST)                                                                                       \
  {                                                                                                            \
    const long tile_ = t_first + ((ST) / 3) * t_stride;                                                        \
    const int dy_ = static_cast<int>((ST) % 3);                                                                \
    const int tx_ = static_cast<int>(tile_ % TX), ty_ = static_cast<int>((tile_ / TX) % TY);                   \
    const int b_ = static_cast<int>(tile_ / (static_cast<long>(TX) * TY));                                     \
    const int iy_ = 2 * (ty_ * 4 + (l32 >> 3)) + dy_ - 1, ix0_ = 2 * (tx_ * 8 + (l32 & 7)) - 1;                \
    const uint16_t* xb_ = x + static_cast<long>(b_) * H * W * CI + half * 8;                                   \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                                         \
      const int ix_ = ix0_ + dx;                                                                               \
      const bool ok_ = iy_ >= 0 && iy_ < H && ix_ >= 0 && ix_ < W;                                             \
      /* (a select AFTER the load would make the compiler wait for it right there: out-of-image taps read a zero page) */ \
      const uint16_t* p_ = ok_ ? xb_ + (static_cast<long>(iy_) * W + ix_) * CI : zero_page;                    \
      const int cstep_ = ok_ ? 16 : 0;                                                                         \
      _Pragma("unroll") for (int cg = 0; cg < CG; ++cg)                                                        \
        DST[dx * CG + cg] = *reinterpret_cast<const bf16x8*>(p_ + cg * cstep_);                                \
    }                                                                                                          \
  }
  f32x16 acc[NB];
  // one row step: prefetch the next one, 27 MFMAs on CUR, tile epilogue after filter row 2
#define C2_STEP(CUR, NXT, ST)                                                                                  \
  {                                                                                                            \
    if ((ST) + 1 < n_steps) C2_LOAD(NXT, (ST) + 1)                                                             \
    const int dy_c = static_cast<int>((ST) % 3);                                                               \
    if (dy_c == 0) {                                                                                           \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;                                       \
    }                                                                                                          \
    _Pragma("unroll") for (int jf = 0; jf < RF; ++jf)                                                          \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                      \
        const bf16x8 wfr = *reinterpret_cast<const bf16x8*>(wl + ((dy_c * RF + jf) * NB + nb) * 1024);         \
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, CUR[jf], acc[nb], 0, 0, 0);                     \
      }                                                                                                        \
    if (dy_c == 2) {                                                                                           \
      /* acc[nb][r]: channel nb*32 + (r & 3) + 8 (r >> 2) + 4 half of position l32 */                          \
      __builtin_amdgcn_wave_barrier();                                                                         \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                        \
        _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                     \
          const int c0 = nb * 32 + 8 * g4 + 4 * half;                                                          \
          const float4 bb = *reinterpret_cast<const float4*>(bias_s + c0);                                     \
          const uint32_t lo = static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 0] + bb.x)) |                     \
                              (static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 1] + bb.y)) << 16);              \
          const uint32_t hi = static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 2] + bb.z)) |                     \
                              (static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 3] + bb.w)) << 16);              \
          *reinterpret_cast<uint2*>(scr + (l32 * CO + c0) * 2) = make_uint2(lo, hi);                           \
        }                                                                                                      \
      __builtin_amdgcn_wave_barrier();                                                                         \
      const long tile_c = t_first + ((ST) / 3) * t_stride;                                                     \
      const int tx_c = static_cast<int>(tile_c % TX), ty_c = static_cast<int>((tile_c / TX) % TY);             \
      const int b_c = static_cast<int>(tile_c / (static_cast<long>(TX) * TY));                                 \
      unsigned char* ob_ = reinterpret_cast<unsigned char*>(out) +                                             \
                           ((static_cast<long>(b_c) * OH + ty_c * 4) * OW + tx_c * 8) * CO * 2;                \
      constexpr int ROWB = 8 * CO * 2;                      /* bytes of 8 positions: contiguous in the NHWC output */ \
      _Pragma("unroll") for (int q = 0; q < TILE_BYTES / 1024; ++q) {                                          \
        const int off = (q * 64 + lane) * 16, ry = off / ROWB, cb = off - ry * ROWB;                           \
        *reinterpret_cast<uint4*>(ob_ + static_cast<long>(ry) * OW * CO * 2 + cb) = *reinterpret_cast<const uint4*>(scr + off); \
      }                                                                                                        \
    }                                                                                                          \
  }
  bf16x8 fa[RF], fb[RF];
  if (n_steps > 0) C2_LOAD(fa, 0)
  for (long st = 0; st < n_steps; st += 2) {
    C2_STEP(fa, fb, st)
    if (st + 1 < n_steps) C2_STEP(fb, fa, st + 1)
  }
#undef C2_STEP
#undef C2_LOAD
}

// ------------------------------------------------------------------ input gradient
template <int CI, int CO>
__global__ __launch_bounds__(kWaves * 64) void conv2_dgrad_kernel(const uint16_t* __restrict__ dy_, const uint16_t* __restrict__ wd,
                                                                  uint16_t* __restrict__ dx_, int N, int H, int W) {
  constexpr int KSD = CO / 16, NB = (CI + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 9 * KSD * NB * 64; i += kWaves * 64) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(wd)[i];
  __syncthreads();
  const unsigned char* wl = lds + lane * 16;
  const uint16_t* zero_page = wd + 9 * KSD * NB * 512;
  const int OH = H / 2, OW = W / 2;                   // dy is [N, OH, OW, CO]; a parity class of dx is an OH x OW grid
  const int TX = OW / 8, TY = OH / 4;
  const long per_class = static_cast<long>(N) * TY * TX, n_tiles = 4 * per_class;
  for (long t = static_cast<long>(blockIdx.x) * kWaves + wave; t < n_tiles; t += static_cast<long>(gridDim.x) * kWaves) {
    // classes in the order (odd, odd), (odd, even), (even, odd), (even, even): 4, 2, 2, 1 taps - heavy tiles first
    const int cls = static_cast<int>(t / per_class);
    const long tile = t - cls * per_class;
    const int py = cls < 2 ? 1 : 0, px = (cls & 1) ? 0 : 1;
    const int tx = static_cast<int>(tile % TX), ty = static_cast<int>((tile / TX) % TY), b = static_cast<int>(tile / (static_cast<long>(TX) * TY));
    const int gy = ty * 4 + (l32 >> 3), gx = tx * 8 + (l32 & 7);          // position inside the class grid
    const int iy = 2 * gy + py, ix = 2 * gx + px;
    const uint16_t* db = dy_ + static_cast<long>(b) * OH * OW * CO + half * 8;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    // taps (ky, kx) with iy + 1 - ky even: iy even -> ky = 1; iy odd -> ky in {0, 2};  oy = (iy + 1 - ky) / 2
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (a == 1 && py == 0) break;
      const int ky = py ? 2 * a : 1;
      const int oy = (iy + 1 - ky) >> 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c == 1 && px == 0) break;
        const int kx = px ? 2 * c : 1;
        const int ox = (ix + 1 - kx) >> 1;
        const bool ok = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
        const uint16_t* p = ok ? db + (static_cast<long>(oy) * OW + ox) * CO : zero_page;
        const int kstep = ok ? 16 : 0;
        const int tap = ky * 3 + kx;
        bf16x8 af[KSD];
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(p + ks * kstep);
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(wl + ((tap * KSD + ks) * NB + nb) * 1024);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfrag, acc[nb], 0, 0, 0);
          }
      }
    }
    uint16_t* ob = dx_ + static_cast<long>(b) * H * W * CI;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
      const long pos = static_cast<long>(2 * (ty * 4 + (i >> 3)) + py) * W + 2 * (tx * 8 + (i & 7)) + px;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int ci = nb * 32 + l32;
        if (ci < CI) ob[pos * CI + ci] = to_bf16(acc[nb][r]);
      }
    }
  }
}

template <int CI, int CO>
int launch_fwd(const uint16_t* x, const uint16_t* wf, const float* bias, uint16_t* out, int N, int H, int W, hipStream_t s) {
  constexpr int kFwdWaves = FwdWaves<CI>::value;
  constexpr int LDS = 9 * (CI / 16) * (CO / 32) * 1024 + CO * 4 + kFwdWaves * 32 * CO * 2;
  static_assert(LDS <= 160 * 1024 && CO == 96, "filter + per-wavefront output tiles must fit the LDS");
  auto kfn = conv2_fwd_kernel<CI, CO>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const long n_tiles = static_cast<long>(N) * (H / 8) * (W / 16);
  const int grid = static_cast<int>(n_tiles < 256L * kFwdWaves ? (n_tiles + kFwdWaves - 1) / kFwdWaves : 256);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(kFwdWaves * 64), LDS, s, x, wf, bias, out, N, H, W);
  return launch_status();
}

template <int CI, int CO>
int launch_dgrad(const uint16_t* dy, const uint16_t* wd, uint16_t* dx, int N, int H, int W, hipStream_t s) {
  constexpr int LDS = 9 * (CO / 16) * ((CI + 31) / 32) * 1024;
  auto kfn = conv2_dgrad_kernel<CI, CO>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const long n_tiles = 4L * N * (H / 8) * (W / 16);
  const int grid = static_cast<int>(n_tiles < 256L * kWaves ? (n_tiles + kWaves - 1) / kWaves : 256);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves * 64), LDS, s, dy, wd, dx, N, H, W);
  return launch_status();
}

}  // namespace

extern "C" {

int cnx_conv3x3s2_supported(int32_t CI, int32_t CO, int32_t H, int32_t W) {
  const bool geo = H > 0 && W > 0 && H % 8 == 0 && W % 16 == 0;            // 4 x 8 output tiles
  return (geo && CO == 96 && (CI == 48 || CI == 64)) ? 1 : 0;
}

int64_t cnx_conv3x3s2_packed_elems(int32_t CI, int32_t CO) {
  return 9L * (CI / 16) * (CO / 32) * 512 + 9L * (CO / 16) * ((CI + 31) / 32) * 512 + 64;   // + a zero page for taps outside the image
}

int cnx_conv3x3s2_pack(const void* w, int w_dtype, void* packed, int32_t CI, int32_t CO, void* stream) {
  if (!w || !packed) return APGD_ERR_NULL;
  if (CI <= 0 || CO <= 0 || CI % 16 != 0 || CO % 32 != 0) return APGD_ERR_SIZE;
  const long nf = 9L * (CI / 16) * (CO / 32) * 512, total = cnx_conv3x3s2_packed_elems(CI, CO);
  uint16_t* wf = static_cast<uint16_t*>(packed);
  const dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
  if (w_dtype == APGD_F32)
    hipLaunchKernelGGL(conv2_pack_kernel<float>, grid, block, 0, as_stream(stream), static_cast<const float*>(w), wf, wf + nf, CI, CO);
  else if (w_dtype == APGD_BF16)
    hipLaunchKernelGGL(conv2_pack_kernel<__bf16>, grid, block, 0, as_stream(stream), static_cast<const __bf16*>(w), wf, wf + nf, CI, CO);
  else return APGD_ERR_DTYPE;
  return launch_status();
}

int cnx_conv3x3s2_fwd(const void* x, const void* packed, const float* bias, void* out, int64_t N, int32_t H, int32_t W, int32_t CI,
                      int32_t CO, void* stream) {
  if (N < 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!x || !packed || !out) return APGD_ERR_NULL;
  if (!cnx_conv3x3s2_supported(CI, CO, H, W) || N > 0x7fffffff) return APGD_ERR_ARG;
  const auto* xp = static_cast<const uint16_t*>(x);
  const auto* wf = static_cast<const uint16_t*>(packed);
  auto* o = static_cast<uint16_t*>(out);
  if (CI == 48) return launch_fwd<48, 96>(xp, wf, bias, o, static_cast<int>(N), H, W, as_stream(stream));
  return launch_fwd<64, 96>(xp, wf, bias, o, static_cast<int>(N), H, W, as_stream(stream));
}

int cnx_conv3x3s2_dgrad(const void* dy, const void* packed, void* dx, int64_t N, int32_t H, int32_t W, int32_t CI, int32_t CO,
                        void* stream) {
  if (N < 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!dy || !packed || !dx) return APGD_ERR_NULL;
  if (!cnx_conv3x3s2_supported(CI, CO, H, W) || N > 0x7fffffff) return APGD_ERR_ARG;
  const auto* wd = static_cast<const uint16_t*>(packed) + 9L * (CI / 16) * (CO / 32) * 512;
  const auto* g = static_cast<const uint16_t*>(dy);
  auto* o = static_cast<uint16_t*>(dx);
  if (CI == 48) return launch_dgrad<48, 96>(g, wd, o, static_cast<int>(N), H, W, as_stream(stream));
  return launch_dgrad<64, 96>(g, wd, o, static_cast<int>(N), H, W, as_stream(stream));
}

}  // extern "C"
