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
//   dgrad     a wavefront = a 4 x 8 block of 2 x 2 input patches against their 2 x 2 output neighbours (see conv2_dgrad_kernel).
// One workgroup of 6 - 12 wavefronts per CU shares the filter; wavefronts take tiles in a grid-stride loop.
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


// ------------------------------------------------------------------ filter pre-arrangement
// forward  piece (ks, nb), ks = tap * CI/16 + cg: lane (n = l32, half), e -> W[nb*32 + n][cg*16 + half*8 + e][tap/3][tap%3]
// (the input gradient's pieces: see conv2_pack_dgrad_kernel below; packed buffer = [forward pieces | 64 zeros | dgrad pieces | 64 zeros])
template <typename TW>
__global__ __launch_bounds__(256) void conv2_pack_kernel(const TW* __restrict__ w, uint16_t* __restrict__ wf, int CI, int CO) {
  const int KSF = 9 * (CI / 16), NBF = CO / 32;
  const long nf = static_cast<long>(KSF) * NBF * 512;
  const long q = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= nf + 64) return;
  if (q >= nf) { wf[q] = 0; return; }                                  // the zero page behind the pieces (taps outside the image)
  const int e = q & 7, lane = (q >> 3) & 63, piece = static_cast<int>(q >> 9);
  const int nb = piece % NBF, ks = piece / NBF, tap = ks / (CI / 16), cg = ks % (CI / 16);
  const int co = nb * 32 + (lane & 31), ci = cg * 16 + (lane >> 5) * 8 + e;
  wf[q] = to_bf16(static_cast<float>(w[(static_cast<long>(co) * CI + ci) * 9 + tap]));
}

// ------------------------------------------------------------------ forward
// Per wavefront a stream of "row steps" (tile, dy): the 3 * CI/16 operand fragments of filter row dy are loaded one step ahead of the
// MFMAs that use them (two register sets), so the HBM / L2 latency of a step hides behind the previous step's 9 * CI/16 * CO/32
// MFMAs.  D[co][pos] (filter fragment first): a lane owns one position and, per accumulator register group, 4 consecutive channels
// - 8-byte writes into a wavefront-private LDS tile, which leaves as 16 bytes per lane of contiguous NHWC rows.
// (wavefronts per workgroup: C2_MAX_WAVES, see above)
// one LDS fragment read per MFMA in the schedule: left alone the scheduler hoists a dozen filter-fragment reads ahead of the MFMAs
// and the 12-wavefront builds (168 registers) spill loop invariants, whose reloads then queue behind the prefetched loads
#define C2_PACE { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
// Wavefronts per workgroup.  Round 2 ran 12 (forward) / 11 (input gradient) at CI = 48 - three per SIMD, 168 registers - and both
// kernels spilled 4-10 registers whose reloads queue behind the prefetched operand loads; with 8 (two per SIMD, no scratch) the
// same box measured 133-151 vs 140-149 us forward and 152-173 vs 150-167 us input gradient, i.e. no difference.  8 it is.
#ifndef C2_MAX_WAVES
#define C2_MAX_WAVES 8
#endif
template <int CI> struct FwdWaves { static constexpr int value = (9 * (CI / 16) * 3 + 12 * 6 + 1 <= 160 && C2_MAX_WAVES >= 12) ? 12 : 8; };

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
  const uint16_t* zero_page = wf + KS * NB * 512;            // 64 zero elements behind the forward pieces
  const int OH = H / 2, OW = W / 2, TX = OW / 8, TY = OH / 4;
  // tile / step counters are 32-bit (the C ABI refuses more than 2^29 tiles): 64-bit wave-uniform divisions and counters cost the
  // 12-wavefront build (168 registers) five spilled registers, reloaded between the prefetched loads of every step
  const int n_tiles = N * TY * TX;
  const int t_first = static_cast<int>(blockIdx.x) * kFwdWaves + wave, t_stride = static_cast<int>(gridDim.x) * kFwdWaves;
  const int my_tiles = t_first < n_tiles ? (n_tiles - t_first + t_stride - 1) / t_stride : 0;
  const int n_steps = 3 * my_tiles;

  // operand fragments of row step `st` -> DST
#define C2_LOAD(DST, ST)                                                                                       \
  {                                                                                                            \
    const int tile_ = t_first + ((ST) / 3) * t_stride;                                                         \
    const int dy_ = (ST) % 3;                                                                                  \
    const int tx_ = tile_ % TX, ty_ = (tile_ / TX) % TY;                                                       \
    const int b_ = tile_ / (TX * TY);                                                                          \
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
    const int dy_c = (ST) % 3;                                                                                 \
    if (dy_c == 0) {                                                                                           \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;                                       \
    }                                                                                                          \
    _Pragma("unroll") for (int jf = 0; jf < RF; ++jf)                                                          \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                      \
        const bf16x8 wfr = *reinterpret_cast<const bf16x8*>(wl + ((dy_c * RF + jf) * NB + nb) * 1024);         \
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, CUR[jf], acc[nb], 0, 0, 0);                     \
        C2_PACE                                                                                                \
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
      const int tile_c = t_first + ((ST) / 3) * t_stride;                                                      \
      const int tx_c = tile_c % TX, ty_c = (tile_c / TX) % TY;                                                 \
      const int b_c = tile_c / (TX * TY);                                                                      \
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
  for (int st = 0; st < n_steps; st += 2) {
    C2_STEP(fa, fb, st)
    if (st + 1 < n_steps) C2_STEP(fb, fa, st + 1)
  }
#undef C2_STEP
#undef C2_LOAD
}

// ------------------------------------------------------------------ input gradient
// A wavefront = a 4 x 8 block of 2 x 2 input PATCHES (128 input positions).  With stride 2 the four positions of patch (gy, gx) -
// classes (py, px) - see exactly the 2 x 2 output positions q = (a, b): (gy + a, gx + b), each through at most one tap:
//     a = 0, py = 0 -> ky = 1;   a = 0, py = 1 -> ky = 2;   a = 1, py = 1 -> ky = 0;   (a = 1, py = 0: none)       same in x
// so  D[(class, ci)][patch] = sum over (q, co) of  Wp[(class, ci)][(q, co)] dY[(q, co)][patch]  with 9 of the 16 (q, class) blocks
// non-zero.  K = 4 CO in 4 steps of CO/16 fragments (one neighbour q per step, its dY row read as 16-byte loads, one step ahead of
// the MFMAs), N = 4 CI in 32-wide blocks of which only those touching a connected class are computed (15 of 24 at CI = 48).
// The result leaves through a wavefront-private LDS tile, one row parity at a time: the two positions of a patch in a row are
// adjacent in memory, so a row of 8 patches is one contiguous run.
__host__ __device__ constexpr bool dg_connected(int q, int cls) {
  const int a = q >> 1, b = q & 1, py = cls >> 1, px = cls & 1;
  return (a == 0 || py == 1) && (b == 0 || px == 1);
}
__host__ __device__ constexpr bool dg_block_used(int q, int nb, int CI) {      // does N-block nb touch a class connected to q?
  for (int cls = 0; cls < 4; ++cls)
    if (dg_connected(q, cls) && nb * 32 < (cls + 1) * CI && (nb + 1) * 32 > cls * CI) return true;
  return false;
}
__host__ __device__ constexpr int dg_blocks_before(int q, int nb, int CI) {    // used (q', nb') pairs before (q, nb), q-major
  int n = 0;
  for (int qq = 0; qq <= q; ++qq)
    for (int b = 0; b < 4 * CI / 32; ++b) {
      if (qq == q && b >= nb) break;
      if (dg_block_used(qq, b, CI)) ++n;
    }
  return n;
}
__host__ __device__ constexpr int dg_tap(int q, int cls) {                      // ky * 3 + kx of the tap connecting them
  const int a = q >> 1, b = q & 1, py = cls >> 1, px = cls & 1;
  const int ky = a ? 0 : (py ? 2 : 1), kx = b ? 0 : (px ? 2 : 1);
  return ky * 3 + kx;
}

// dgrad filter pieces: for q, for used nb (in that order), for ks: piece index = (dg_blocks_before(q, nb) * KSD + ks);
// lane (n = l32, half), e -> Wp[nb*32 + n][(q, ks*16 + half*8 + e)]
template <typename TW>
__global__ __launch_bounds__(256) void conv2_pack_dgrad_kernel(const TW* __restrict__ w, uint16_t* __restrict__ wd, int CI, int CO) {
  const int KSD = CO / 16, NBLK = 4 * CI / 32;
  const int n_used = dg_blocks_before(3, NBLK, CI);
  const long total = static_cast<long>(n_used) * KSD * 512;
  const long p = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= total + 64) return;
  if (p >= total) { wd[p] = 0; return; }                               // the zero page behind the pieces
  const int e = p & 7, lane = (p >> 3) & 63, piece = static_cast<int>(p >> 9);
  const int ks = piece % KSD, ub = piece / KSD;
  int q = 0, nb = 0;
  for (int qq = 0; qq < 4; ++qq)
    for (int b = 0; b < NBLK; ++b)
      if (dg_block_used(qq, b, CI) && dg_blocks_before(qq, b, CI) == ub) { q = qq; nb = b; }
  const int n = nb * 32 + (lane & 31), cls = n / CI, ci = n - cls * CI;
  const int co = ks * 16 + (lane >> 5) * 8 + e;
  float v = 0.f;
  if (dg_connected(q, cls)) v = static_cast<float>(w[(static_cast<long>(co) * CI + ci) * 9 + dg_tap(q, cls)]);
  wd[p] = to_bf16(v);
}

template <int CI> struct DgWaves {
  static constexpr int WKB = dg_blocks_before(3, 4 * CI / 32, CI) * 6;          // KiB of filter pieces (CO = 96: 6 k-steps)
  static constexpr int SKB = 32 * 2 * CI * 2 / 1024;                            // KiB of one wavefront's half-tile
  static constexpr int value = (158 - WKB) / SKB > C2_MAX_WAVES ? C2_MAX_WAVES : (158 - WKB) / SKB;
};

template <int CI, int CO, int NW = DgWaves<CI>::value>
__global__ __launch_bounds__(NW * 64) void conv2_dgrad_kernel(const uint16_t* __restrict__ dy_, const uint16_t* __restrict__ wd,
                                                              uint16_t* __restrict__ dx_, int N, int H, int W) {
  constexpr int KSD = CO / 16, NBLK = 4 * CI / 32, NUSED = dg_blocks_before(3, NBLK, CI);
  constexpr int WBYTES = NUSED * KSD * 1024, HALF_BYTES = 32 * 2 * CI * 2;
  static_assert((2 * CI) % 32 == 0 && CI % 4 == 0, "a row parity is a whole number of N-blocks; 4 consecutive n share a class");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* scr = lds + WBYTES + wave * HALF_BYTES;
  for (int i = tid; i < NUSED * KSD * 64; i += NW * 64) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(wd)[i];
  __syncthreads();
  const unsigned char* wl = lds + lane * 16;
  const uint16_t* zero_page = wd + NUSED * KSD * 512;
  const int OH = H / 2, OW = W / 2, TX = OW / 8, TY = OH / 4;
  const int n_tiles = N * TY * TX;                      // 32-bit counters: see conv2_fwd_kernel
  const int t_first = static_cast<int>(blockIdx.x) * NW + wave, t_stride = static_cast<int>(gridDim.x) * NW;
  const int my_tiles = t_first < n_tiles ? (n_tiles - t_first + t_stride - 1) / t_stride : 0;
  const int n_steps = 4 * my_tiles;
#define DG_LOAD(DST, ST)                                                                                       \
  {                                                                                                            \
    const int tile_ = t_first + ((ST) >> 2) * t_stride;                                                        \
    const int q_ = (ST) & 3;                                                                                   \
    const int tx_ = tile_ % TX, ty_ = (tile_ / TX) % TY;                                                       \
    const int b_ = tile_ / (TX * TY);                                                                          \
    const int oy_ = ty_ * 4 + (l32 >> 3) + (q_ >> 1), ox_ = tx_ * 8 + (l32 & 7) + (q_ & 1);                    \
    const bool ok_ = oy_ < OH && ox_ < OW;                                                                     \
    const uint16_t* p_ = ok_ ? dy_ + ((static_cast<long>(b_) * OH + oy_) * OW + ox_) * CO + half * 8 : zero_page; \
    const int kstep_ = ok_ ? 16 : 0;                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < KSD; ++ks) DST[ks] = *reinterpret_cast<const bf16x8*>(p_ + ks * kstep_); \
  }
  f32x16 acc[NBLK];
#define DG_MFMAS(CUR, Q)                                                                                       \
  _Pragma("unroll") for (int nb = 0; nb < NBLK; ++nb)                                                          \
    if (dg_block_used(Q, nb, CI)) {                                                                            \
      _Pragma("unroll") for (int ks = 0; ks < KSD; ++ks) {                                                     \
        const bf16x8 wfr = *reinterpret_cast<const bf16x8*>(wl + (dg_blocks_before(Q, nb, CI) * KSD + ks) * 1024); \
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, CUR[ks], acc[nb], 0, 0, 0);                     \
        C2_PACE                                                                                                \
      }                                                                                                        \
    }
#define DG_STEP(CUR, NXT, ST, Q)                                                                               \
  {                                                                                                            \
    if ((ST) + 1 < n_steps) DG_LOAD(NXT, (ST) + 1)                                                             \
    if ((Q) == 0) {                                                                                            \
      _Pragma("unroll") for (int nb = 0; nb < NBLK; ++nb)                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;                                       \
    }                                                                                                          \
    DG_MFMAS(CUR, Q)                                                                                           \
  }
  bf16x8 fa[KSD], fb[KSD];
  if (n_steps > 0) DG_LOAD(fa, 0)
  for (int st = 0; st < n_steps; st += 4) {
    DG_STEP(fa, fb, st, 0)
    DG_STEP(fb, fa, st + 1, 1)
    DG_STEP(fa, fb, st + 2, 2)
    DG_STEP(fb, fa, st + 3, 3)
    // acc[nb][r]: n = nb*32 + (r & 3) + 8 (r >> 2) + 4 half = class * CI + ci, patch l32
    const int tile_c = t_first + (st >> 2) * t_stride;
    const int tx_c = tile_c % TX, ty_c = (tile_c / TX) % TY;
    const int b_c = tile_c / (TX * TY);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int nb = py * (2 * CI / 32); nb < (py + 1) * (2 * CI / 32); ++nb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = nb * 32 + 8 * g4 + 4 * half - py * 2 * CI;             // (px, ci) index inside this row parity
          const uint32_t lo = static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 0])) | (static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 1])) << 16);
          const uint32_t hi = static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 2])) | (static_cast<uint32_t>(to_bf16(acc[nb][4 * g4 + 3])) << 16);
          *reinterpret_cast<uint2*>(scr + (l32 * 2 * CI + n0) * 2) = make_uint2(lo, hi);
        }
      __builtin_amdgcn_wave_barrier();
      constexpr int ROWB = 8 * 2 * CI * 2;                                      // 8 patches x 2 positions x CI channels: contiguous in dx
      unsigned char* ob = reinterpret_cast<unsigned char*>(dx_) +
                          ((static_cast<long>(b_c) * H + 2 * (ty_c * 4) + py) * W + 2 * (tx_c * 8)) * CI * 2;
#pragma unroll
      for (int qq = 0; qq < HALF_BYTES / 1024; ++qq) {
        const int off = (qq * 64 + lane) * 16, ry = off / ROWB, cb = off - ry * ROWB;
        *reinterpret_cast<uint4*>(ob + static_cast<long>(2 * ry) * W * CI * 2 + cb) = *reinterpret_cast<const uint4*>(scr + off);
      }
    }
  }
#undef DG_STEP
#undef DG_MFMAS
#undef DG_LOAD
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
  constexpr int NW = DgWaves<CI>::value;
  constexpr int LDS = dg_blocks_before(3, 4 * CI / 32, CI) * (CO / 16) * 1024 + NW * 32 * 2 * CI * 2;
  static_assert(LDS <= 160 * 1024 && NW >= 4, "filter pieces + per-wavefront half tiles must fit the LDS");
  auto kfn = conv2_dgrad_kernel<CI, CO>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const long n_tiles = static_cast<long>(N) * (H / 8) * (W / 16);
  const int grid = static_cast<int>(n_tiles < 256L * NW ? (n_tiles + NW - 1) / NW : 256);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), LDS, s, dy, wd, dx, N, H, W);
  return launch_status();
}

}  // namespace

extern "C" {

int cnx_conv3x3s2_supported(int32_t CI, int32_t CO, int32_t H, int32_t W) {
  const bool geo = H > 0 && W > 0 && H % 8 == 0 && W % 16 == 0;            // 4 x 8 output tiles
  return (geo && CO == 96 && (CI == 48 || CI == 64)) ? 1 : 0;
}

static long conv2_fwd_elems(int CI, int CO) { return 9L * (CI / 16) * (CO / 32) * 512 + 64; }
static long conv2_dgrad_elems(int CI, int CO) { return static_cast<long>(dg_blocks_before(3, 4 * CI / 32, CI)) * (CO / 16) * 512 + 64; }

int64_t cnx_conv3x3s2_packed_elems(int32_t CI, int32_t CO) { return conv2_fwd_elems(CI, CO) + conv2_dgrad_elems(CI, CO); }

int cnx_conv3x3s2_pack(const void* w, int w_dtype, void* packed, int32_t CI, int32_t CO, void* stream) {
  if (!w || !packed) return APGD_ERR_NULL;
  if (CI <= 0 || CO <= 0 || CI % 16 != 0 || CO % 32 != 0) return APGD_ERR_SIZE;
  const long nf = conv2_fwd_elems(CI, CO), nd = conv2_dgrad_elems(CI, CO);
  uint16_t* wf = static_cast<uint16_t*>(packed);
  const dim3 gf(static_cast<unsigned>((nf + 255) / 256)), gd(static_cast<unsigned>((nd + 255) / 256)), block(256);
  hipStream_t s = as_stream(stream);
  if (w_dtype == APGD_F32) {
    hipLaunchKernelGGL(conv2_pack_kernel<float>, gf, block, 0, s, static_cast<const float*>(w), wf, CI, CO);
    hipLaunchKernelGGL(conv2_pack_dgrad_kernel<float>, gd, block, 0, s, static_cast<const float*>(w), wf + nf, CI, CO);
  } else if (w_dtype == APGD_BF16) {
    hipLaunchKernelGGL(conv2_pack_kernel<__bf16>, gf, block, 0, s, static_cast<const __bf16*>(w), wf, CI, CO);
    hipLaunchKernelGGL(conv2_pack_dgrad_kernel<__bf16>, gd, block, 0, s, static_cast<const __bf16*>(w), wf + nf, CI, CO);
  } else return APGD_ERR_DTYPE;
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
  const auto* wd = static_cast<const uint16_t*>(packed) + conv2_fwd_elems(CI, CO);
  const auto* g = static_cast<const uint16_t*>(dy);
  auto* o = static_cast<uint16_t*>(dx);
  if (CI == 48) return launch_dgrad<48, 96>(g, wd, o, static_cast<int>(N), H, W, as_stream(stream));
  return launch_dgrad<64, 96>(g, wd, o, static_cast<int>(N), H, W, as_stream(stream));
}

}  // extern "C"
