// wgrad_kernels.hip — contractions over the ROW index m of row-major operands, for gfx950 (round 5):
//
//     D[i][j] = sum_m A[m][i] * B[m][j]            A: [M, NA], B: [M, NB] bf16, both row-major; D fp32
//
// This is the shape of every weight gradient of the path (models/convnext.py:42-46 backward, utils_architecture.py:205-211 backward):
// the pointwise convolutions' dW1 = dHpre^T a and dW2 = dO^T H, the head, and - with im2col rows gathered on the fly - the ConvStem
// convolutions' filter gradients, which were the last MIOpen kernels of the step.  Both MFMA operands want, per lane, 8 consecutive
// values of the contraction index for ONE row / column of D - i.e. a column of the row-major operand.  gfx950's LDS transpose read
// does exactly that on the way out of LDS: `ds_read_b64_tr_b16` takes, per group of 16 lanes, sixteen 8-byte pieces (lane i of the
// group: 4 consecutive bf16 of row i / 4 at column chunk i % 4 of a 4 x 16 block - each lane hands over its own 8-byte-aligned
// address, so the row stride is free) and returns to lane i COLUMN i of the block: four consecutive m for one column.  Two reads
// make one operand fragment of `mfma_f32_32x32x16_bf16` (lane l: row / column l % 32, k = 8 (l / 32) .. + 7): the operand tiles
// go from HBM to LDS as they lie in memory (16-byte pieces), no transposing pass over either operand exists anywhere.
// (/opt/skills/guides/cdna_hip_programming.md T10; tools/probe/tr_probe.cpp prints the lane map this file relies on.)
//
// Kernels:
//   stem_conv_wgrad_kernel<P>   filter / bias gradient of Conv2d(3, P, 3, stride 2, padding 1) on the fp32 NCHW image (the attack's
//                               iterate): D[(ci,kh,kw) | 1][p] = sum_pos patch[pos][.] dy[pos][p]; the patch values are gathered
//                               straight from the image (rounded to bf16 as the autocast convolution does), dy rows through LDS
//   wgrad_reduce_kernel         fixed-order sum of the workgroups' partial results (deterministic)
//   conv2_wgrad_kernel<CI,CO>   filter / bias gradient of the second ConvStem convolution (3x3, stride 2) as an implicit GEMM over the positions
//   gemm_tn_kernel<...>         cnx_gemm_tn / _ex: D = A^T B over the rows of two row-major (or accumulator-order tile) operands - every
//                               pointwise weight gradient; PAIR: both weight gradients of one block in one launch (cnx_gemm_tn_pair, round 6);
//                               stage loops: two 64-row buffers (hand-over at the stage boundary or one k-step early), ring of 32-row stages
//   gemm_tn_reduce_kernel, gemm_tn_pair_reduce_kernel   fixed-order sums of the split partials (the pair's also transposes dW2 back)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

#ifndef TN_ABLATE
#define TN_ABLATE 0
#endif

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// LDS byte address of a __shared__ object (the DS instructions take 32-bit LDS addresses)
template <typename T>
__device__ __forceinline__ uint32_t lds_addr(const T* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p));
}

// One transpose read: 4 consecutive contraction indices (rows of the LDS image) for this lane's column.  `addr` is THIS lane's
// piece: 8-byte aligned (an address off by 2 / 4 / 6 silently returns the aligned piece: guide, guideline 17).
__device__ __forceinline__ u32x2 tr_read(uint32_t addr) {
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      reinterpret_cast<s16x4 __attribute__((address_space(3)))*>(static_cast<uintptr_t>(addr)));
  return __builtin_bit_cast(u32x2, r);
}
// Operand fragment of mfma_f32_32x32x16_bf16 from a row-major LDS image: lane l = (k-block kb = l / 32, column c = l % 32) gets
// rows 8 kb .. 8 kb + 7 of column c.  piece = the lane's own 8-byte piece of read 0 (row 8 kb + (l % 16) / 4, columns
// 16 ((l / 16) % 2) + 4 (l % 4) ..); read 1 is four rows further down (row_bytes = the image's row stride in bytes).
__device__ __forceinline__ bf16x8 tr_fragment(uint32_t piece, uint32_t row_bytes) {
  const u32x2 lo = tr_read(piece), hi = tr_read(piece + 4u * row_bytes);
  const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, v);
}
// byte offset of a lane's piece inside a [rows][row_bytes] image for the fragment of columns c0 .. c0 + 31, rows r0 .. r0 + 15
__device__ __forceinline__ uint32_t tr_piece(int lane, int r0, int c0, uint32_t row_bytes) {
  const int g = lane >> 4, i = lane & 15;
  return static_cast<uint32_t>(r0 + 8 * (g >> 1) + (i >> 2)) * row_bytes + static_cast<uint32_t>(c0 + 16 * (g & 1) + 4 * (i & 3)) * 2u;
}

// ------------------------------------------------------------------------------------------------------------------------
// Stem convolution filter gradient.  A wavefront takes chunks of 16 consecutive output positions (flattened (n, oh, ow)):
//   A fragment [32 x 16]: row i = (ci*3 + kh)*3 + kw for i < 27, row 27 = ones (its products are the bias gradient), rest zero;
//                         lane (i, kb) gathers its 8 positions' image values with scalar fp32 loads (out of the image: zero)
//   B fragments [16 x 32] per 32-channel block of dy: the chunk's dy rows are 16 P contiguous bf16 - copied to LDS in 16-byte
//                         pieces and read back transposed
//   acc[nb] += A x B[nb]                                 (27 x P products per position on the matrix pipe)
// ROW16: OW is a multiple of 16 - a chunk lies inside one output row, (n, oh) are wave-uniform and ow needs no wrap; otherwise
// every element derives its own (n, oh, ow) (widths the ConvStem meets only at odd evaluation resolutions).
// The workgroup's four partial results are added through LDS and written to ws[blockIdx.x][28][P].
template <int P, bool ROW16>
__global__ __launch_bounds__(256) void stem_conv_wgrad_kernel(const float* __restrict__ x, const uint16_t* __restrict__ dy,
                                                              float* __restrict__ ws, long total, int N, int H, int W, int OH, int OW) {
  constexpr int NB = (P + 31) / 32;
  constexpr uint32_t ROWB = P * 2;                                   // dy row in bytes
  constexpr int PIECES = 16 * P * 2 / 16;                            // 16-byte pieces of a chunk's dy rows (P = 48: 96)
  constexpr int NLD = (PIECES + 63) / 64;
  __shared__ __attribute__((aligned(16))) uint16_t img[4][2][16 * P + 32 * 8];   // per wave, two chunks; tail: columns P .. 32 NB of the last block read zeros / finite junk
  __shared__ float red[4][28][NB * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kb = lane >> 5;
  const int ci = i / 9, kh = (i - ci * 9) / 3, kw = i - ci * 9 - kh * 3;
  const bool tap = i < 27, one = i == 27;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0,
      static_cast<uint32_t>(static_cast<long>(N) * 3 * H * W * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dy), 0,
      static_cast<uint32_t>(total * P * 2), 0x00020000);
  // zero the image tails once (columns beyond P of the last block are multiplied into rows of D nobody reads, but must be finite)
  {
    uint16_t* flat = &img[wave][0][0];
    for (int e = lane; e < 2 * (16 * P + 32 * 8); e += 64) flat[e] = 0;
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const long nchunks = (total + 15) / 16;
  const long wstride = static_cast<long>(gridDim.x) * 4;
  float xv[8];
  u32x4 dv[NLD];
  auto fetch = [&](long c) {                                          // global -> registers: chunk c's patch values and dy pieces
    const long p0 = c * 16;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int piece = k * 64 + lane;
      const uint32_t off = piece < PIECES ? static_cast<uint32_t>(p0 * ROWB) + static_cast<uint32_t>(piece) * 16u : 0xffffffffu;
      dv[k] = __builtin_amdgcn_raw_buffer_load_b128(rsd, off, 0, 0);  // beyond the tensor (tail chunk): zeros
    }
    if (ROW16) {
      const long r = p0 / OW;                                         // wave-uniform
      const int ow0 = static_cast<int>(p0 - r * OW) + 8 * kb;
      const int oh = static_cast<int>(r % OH);
      const long n = r / OH;
      const int ih = 2 * oh - 1 + kh;
      const bool rok = tap && ih >= 0 && p0 < total;
      const long base = (n * 3 + ci) * static_cast<long>(H) * W + static_cast<long>(ih) * W + (2 * ow0 - 1 + kw);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bool ok = rok && (2 * (ow0 + t) - 1 + kw) >= 0;
        const uint32_t off = ok ? static_cast<uint32_t>(base + 2 * t) * 4u : 0xffffffffu;
        xv[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, off, 0, 0));
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const long p = p0 + 8 * kb + t;
        const long r = p / OW;
        const int ow = static_cast<int>(p - r * OW), oh = static_cast<int>(r % OH);
        const long n = r / OH;
        const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
        const bool ok = tap && p < total && ih >= 0 && iw >= 0;
        const uint32_t off = ok ? static_cast<uint32_t>((n * 3 + ci) * static_cast<long>(H) * W + static_cast<long>(ih) * W + iw) * 4u : 0xffffffffu;
        xv[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, off, 0, 0));
      }
    }
  };

  long c = static_cast<long>(blockIdx.x) * 4 + wave;
  int buf = 0;
  if (c < nchunks) fetch(c);
  for (; c < nchunks; c += wstride) {
    // this chunk: registers -> A fragment / LDS image
    const long p0 = c * 16;
    u32x4 av;
    {
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = one ? ((p0 + 8 * kb + t) < total ? 1.0f : 0.0f) : xv[t];
      av = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    }
    uint16_t* im = img[wave][buf];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int piece = k * 64 + lane;
      if (piece < PIECES) *reinterpret_cast<u32x4*>(im + piece * 8) = dv[k];
    }
    const long cn = c + wstride;
    if (cn < nchunks) fetch(cn);                                      // the next chunk's loads fly under this chunk's LDS round trip + MFMAs
    const bf16x8 a = __builtin_bit_cast(bf16x8, av);
    const uint32_t im_a = lds_addr(im);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const bf16x8 b = tr_fragment(im_a + tr_piece(lane, 0, nb * 32, ROWB), ROWB);
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
    }
    buf ^= 1;                                                        // (the wave's own DS operations execute in order: no barrier)
  }

  // workgroup partial: rows 0 .. 27 of D
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (row < 28) red[wave][row][nb * 32 + i] = acc[nb][r];
    }
  __syncthreads();
  float* out = ws + static_cast<long>(blockIdx.x) * 28 * P;
  for (int e = threadIdx.x; e < 28 * P; e += 256) {
    const int row = e / P, col = e - row * P;
    out[e] = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
  }
}

// LDS-DMA: 16 bytes per lane from a raw buffer straight into LDS; lds_dst is wave-uniform (M0), lane l lands at lds_dst + 16 l.
// Honours EXEC; the compiler sees neither the load nor its LDS write: completion is the caller's s_waitcnt vmcnt.
__device__ __forceinline__ void dma_lds16(uint32_t lds_dst, uint32_t voff, u32x4 rs, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc4(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  return u32x4{static_cast<uint32_t>(a), static_cast<uint32_t>(a >> 32) & 0xffffu, bytes, 0x00020000u};   // stride 0: raw buffer
}

// ------------------------------------------------------------------------------------------------------------------------
// Filter / bias gradient of the second ConvStem convolution, Conv2d(CI, CO, 3, stride 2, padding 1) on channels-last bf16 rows
// (utils_architecture.py:205-211: 48 -> 96; ConvBlock3(64): 64 -> 96):
//     dW[co][(kh, kw, ci)] = sum_{n, oh, ow} dy[n, oh, ow, co] * x[n, 2 oh - 1 + kh, 2 ow - 1 + kw, ci]       (66.6 GFLOP at batch 256)
// An implicit-GEMM contraction over the positions with BOTH operands coming out of LDS through transpose reads.  A workgroup (four
// wavefronts) takes one output row (n, oh) at a time: the three input rows it touches go to LDS as they lie in memory
// ([3][WP][CI] bf16; image column iw sits at slot iw + 1, slot 0 and the slots behind the row stay zero: padding and the k-step
// tail), the dy row likewise ([16 KS][CO], rows >= OW stay zero).  D is [CO = 3 blocks of 32] x [9 CI columns, flat index
// c = (kh*3 + kw)*CI + ci - the weight's own channels-last order - in blocks of 32]; a 16-column group of a block lies inside one tap
// because CI % 16 == 0, so a lane's piece address is  position part (2 ow * CI * 2 bytes) + column part (tap offset + ci), the k-step
// and the second read are immediates.  Column blocks are dealt over the workgroup's eight wavefronts (block cb -> wavefront cb % 8,
// two wavefronts per SIMD sharing the row images); the first wavefront with one block fewer also owns the bias gradient: one more
// "column block" whose B fragment is a constant (column 0 = ones).  Per k-step and wavefront: 6 + 2 NCB transpose reads for 3 NCB
// MFMAs.  One partial result per workgroup, summed in fixed order.
constexpr int kW2 = 8;                                                 // wavefronts per workgroup
template <int CI, int CO, int NCB>
__global__ __launch_bounds__(kW2 * 64, 2) void conv2_wgrad_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                                             float* __restrict__ ws, int N, int H, int W) {
  static_assert(CO == 96 && CI % 16 == 0, "shapes");
  constexpr int KT = 9 * CI;                                           // columns of D
  constexpr int NCOLB = (KT + 31) / 32;                                // 32-column blocks (the last may be half empty)
  constexpr int BIAS_WAVE = NCOLB % kW2;                               // the first wavefront with one block fewer (NCOLB % kW2 != 0 here)
  static_assert(NCOLB % kW2 != 0 && (NCOLB + kW2 - 1) / kW2 == NCB, "column-block deal");
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int OH = H / 2, OW = W / 2, KS = (OW + 15) / 16, WP = 32 * KS + 3;
  uint16_t* ximg = lds;                                                // [3][WP][CI]      (two buffers of both images: a row is
  uint16_t* dimg = lds + 3 * WP * CI;                                  // [16 KS][CO]       loaded while the one before is multiplied)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15;
  for (int e = tid; e < 2 * (3 * WP * CI + 16 * KS * CO) / 8; e += kW2 * 64) reinterpret_cast<u32x4*>(lds)[e] = u32x4{0u, 0u, 0u, 0u};

  // the lane's piece offsets: A (dy image, row stride 2 CO bytes) and the position part of B (a position = 2 image columns)
  const uint32_t a_piece = lds_addr(dimg) + static_cast<uint32_t>(8 * (g >> 1) + (i16 >> 2)) * (CO * 2) + static_cast<uint32_t>(16 * (g & 1) + 4 * (i16 & 3)) * 2u;
  const uint32_t b_pos = lds_addr(ximg) + static_cast<uint32_t>(8 * (g >> 1) + (i16 >> 2)) * (4 * CI);
  uint32_t b_piece[NCB];
#pragma unroll
  for (int j = 0; j < NCB; ++j) {
    int c = 32 * (wave + kW2 * j) + 16 * (g & 1) + 4 * (i16 & 3);
    if (c >= KT) c = 0;                                                // (columns behind the filter: any finite data, never stored)
    const int tap = c / CI, ci = c - tap * CI, kh = tap / 3, kw = tap - 3 * kh;
    b_piece[j] = b_pos + static_cast<uint32_t>((kh * WP + kw) * CI + ci) * 2u;
  }
  const int ncb = (NCOLB - wave + kW2 - 1) / kW2;                      // this wavefront's blocks: wave, wave + 8, ...
  const bool bias_wave = wave == BIAS_WAVE;
  bf16x8 ones;                                                         // B fragment of the bias "block": column 0 = 1
  {
    const uint32_t v = (lane & 31) == 0 ? 0x3f803f80u : 0u;
    const u32x4 t = {v, v, v, v};
    ones = __builtin_bit_cast(bf16x8, t);
  }
  f32x16 acc[3][NCB];
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][j][r] = 0.f;

  const u32x4 rsx = make_rsrc4(x, static_cast<uint32_t>(static_cast<long>(N) * H * W * CI * 2));
  const u32x4 rsd = make_rsrc4(dy, static_cast<uint32_t>(static_cast<long>(N) * OH * OW * CO * 2));
  const int xrow_pieces = W * CI / 8, drow_pieces = OW * CO / 8;       // 16-byte pieces per input / dy row
  const long rows = static_cast<long>(N) * OH;
  const uint32_t buf_elems = static_cast<uint32_t>(3 * WP * CI + 16 * KS * CO);
  // A row's images arrive by LDS-DMA (buffer_load_dwordx4 ... lds: 64 consecutive 16-byte pieces per instruction land at M0 + 16 l,
  // no staging registers): units of 64 pieces - 3 x NXU for the input rows, NDU for the dy row - dealt over the eight wavefronts.
  // Lanes behind a row's last piece are switched off (EXEC): the slots / rows behind the data keep their zeros.  The filter row
  // above the image (oh = 0, kh = 0) is written as zeros by hand.  Completion: s_waitcnt vmcnt(0) in front of the barrier that
  // hands the buffer over (the compiler does not see these loads).
  const int NXU = (xrow_pieces + 63) / 64, NDU = (drow_pieces + 63) / 64;
  const uint32_t lds0 = lds_addr(lds);
  auto dma_row = [&](long row, int buf) {
    const long n = row / OH;
    const int oh = static_cast<int>(row - n * OH);
    const uint32_t xb = lds0 + static_cast<uint32_t>(buf) * buf_elems * 2u, db = xb + static_cast<uint32_t>(3 * WP * CI) * 2u;
    for (int u = wave; u < 3 * NXU + NDU; u += kW2) {                  // wave-uniform
      if (u < 3 * NXU) {
        const int kh = u / NXU, blk = u - kh * NXU, pc = blk * 64 + lane;
        const int ih = 2 * oh - 1 + kh;
        const uint32_t dst = xb + static_cast<uint32_t>((kh * WP + 1) * CI) * 2u + static_cast<uint32_t>(blk) * 1024u;
        if (ih >= 0) {
          if (pc < xrow_pieces) dma_lds16(dst, static_cast<uint32_t>(pc) * 16u, rsx, static_cast<uint32_t>(((n * H + ih) * W) * CI * 2));
        } else if (pc < xrow_pieces) {
          *reinterpret_cast<u32x4 __attribute__((address_space(3)))*>(static_cast<uintptr_t>(dst + static_cast<uint32_t>(lane) * 16u)) = u32x4{0u, 0u, 0u, 0u};
        }
      } else {
        const int blk = u - 3 * NXU, pc = blk * 64 + lane;
        if (pc < drow_pieces) dma_lds16(db + static_cast<uint32_t>(blk) * 1024u, static_cast<uint32_t>(pc) * 16u, rsd, static_cast<uint32_t>(row * OW * CO * 2));
      }
    }
  };
  long row = blockIdx.x;
  int buf = 0;
  __syncthreads();                                                     // zero fill done
  if (row < rows) dma_row(row, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; row < rows; row += gridDim.x) {
    const long next = row + gridDim.x;
    if (next < rows) dma_row(next, buf ^ 1);                           // (that buffer: last read one row ago, behind a barrier)
    const uint32_t boff = static_cast<uint32_t>(buf) * buf_elems * 2u;
    const uint32_t a_cur = a_piece + boff;
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[3];
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) a[cb] = tr_fragment(a_cur + static_cast<uint32_t>(ks * 16 * CO * 2 + cb * 64), CO * 2);
#pragma unroll
      for (int j = 0; j < NCB; ++j) {
        if (j < ncb) {
          const bf16x8 b = tr_fragment(b_piece[j] + boff + static_cast<uint32_t>(ks * 16 * 4 * CI), 4 * CI);
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) acc[cb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb], b, acc[cb][j], 0, 0, 0);
        } else if (j == NCB - 1 && bias_wave) {
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) acc[cb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb], ones, acc[cb][j], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }
  // partial result: ws[blockIdx.x][CO][KT] then [CO] bias sums
  float* out = ws + static_cast<long>(blockIdx.x) * (CO * KT + CO);
  const int col = lane & 31, kb = lane >> 5;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const int c = 32 * (wave + kW2 * j) + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (j < ncb) { if (c < KT) out[co * KT + c] = acc[cb][j][r]; }
        else if (j == NCB - 1 && bias_wave && col == 0) out[CO * KT + co] = acc[cb][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// cnx_gemm_tn:  D[N1][N2] = A^T B,  A [M, N1], B [M, N2] bf16, D fp32 - the weight gradients of the pointwise convolutions / linears
// (dW1 = dHpre^T a, dW2 = dO^T H: models/convnext.py:42-46 backward).
//
// Operand layouts (per operand):
//   ROWS  row-major [M][N] with a leading dimension (activations as every other kernel writes them);
//   ACC   tiles [M / 32][N / 32] of 2 KiB in the order the fused block kernels hold a 32 x 32 tile of the hidden activation in their
//         MFMA accumulators: element (m, n) of a tile at byte 64 m + 32 ((n / 4) % 2) + 8 (n / 8) + 2 (n % 4) - a lane of those kernels
//         (row m = lane % 32, half = lane / 32) owns the 32 bytes at 64 m + 32 half = its 16 accumulator values as bf16, and stores
//         them with two 16-byte instructions, no transposition anywhere (cnx_block_mlp_*: Hpre workspace, emitted H / dHpre).
// Either way 4 consecutive n of one row m are 8 contiguous, 8-byte-aligned bytes: a transpose-read piece.
//
// LDS image of an operand tile: per 32 columns a sub-image of 64 rows (one stage) x 64 bytes.  ROWS: row m at 64 m, the 32 columns
// in order; ACC: the two 2 KiB tiles of the stage verbatim.  A fragment of mfma_f32_32x32x16_bf16 (32 columns x 16 rows) is ONE
// contiguous KiB in both; a transpose read's two 16-lane groups of a half-wave take rows r .. r + 3 x 64 bytes = 256 contiguous
// bytes: every LDS bank once (no padding, no swizzle).  The LDS-DMA instruction that fills such a KiB reads 16 rows x 64 bytes
// (ROWS: lane l = row l / 4, 16-byte piece l % 4) or one contiguous KiB (ACC) and writes lane-linear.
// Workgroup = WI x WJ wavefronts, wavefront tile (32 F) x (32 F): BM = 32 F WI rows of D (columns of A), BN = 32 F WJ columns.
// K runs over M in stages of 64 rows, two stages in LDS: the DMA of stage t + 1 is issued before the 4 k-steps of stage t
// (per k-step and wavefront: 2 F + 2 F transpose reads for F x F MFMAs), s_waitcnt vmcnt(0) + one barrier per stage.  M is split
// over workgroups (rows_per_split, a multiple of 64): every workgroup writes its fp32 partial tile, gemm_tn_reduce_kernel adds the
// splits in a fixed order.  Workgroups of one split are neighbours on an XCD (they re-read each other's operand columns from L2).
// CS: also the column sums of A (= the bias gradient that belongs to this weight gradient: db1 = sum_m dHpre, db2 = sum_m dO) -
// the wavefronts of the first tile column multiply their A fragments with a constant B fragment (column 0 = ones): F more MFMAs
// per k-step there, no pass over A of its own.
constexpr int kRows = 0, kAcc = 1;
template <int LAYOUT>
__device__ __forceinline__ uint32_t tn_lane_off(int lane) {            // the lane's piece inside a fragment's KiB (k-step 0 of a stage, read 0)
  const int g = lane >> 4, i = lane & 15;
  const int row = 8 * (g >> 1) + (i >> 2), hq = 4 * (g & 1) + (i & 3);       // hq: the 4-column chunk of the fragment's 32 columns
  if (LAYOUT == kRows) return static_cast<uint32_t>(row * 64 + hq * 8);
  return static_cast<uint32_t>(row * 64 + (hq & 1) * 32 + (hq >> 1) * 8);
}
// PAIR: TWO contractions of the same shape and row count in one launch - the two weight gradients of one block,
//   problem 0:  D0 [N1][N2] = A^T B,    column sums of A     (dW1 = dHpre^T LN(u), d(b1) = sum dHpre:  A = dHpre tiles, B = LN(u) rows)
//   problem 1:  T  [N1][N2] = A2^T B2,  column sums of B2    (dW2^T = H^T dO,      d(b2) = sum dO:     A2 = H tiles,    B2 = dO rows)
// - dW2 is computed TRANSPOSED so that both problems have the tile operand on the A side and one kernel body serves both (the fixed-
// order sum writes it back as [N2][N1]).  A split's workgroups are the tiles of BOTH problems: half as many splits as two launches
// need to fill the chip, i.e. half the partial results to write and to sum (75 -> 37 MB per contraction at C = 384).
// KT / NBUF: rows per stage and stages in LDS.  64 / 2 (the single contractions): stage t + 1 in flight while stage t is multiplied.
// 32 / 4 (the pair at its 384 x 192 tile, 36 KiB per stage): THREE stages in flight - the stage period of the two-buffer loop
// (2.5 us per 64 rows at C = 384) was the round trip of one stage's DMA, not its MFMA time (1.1 us): s_waitcnt vmcnt(n) leaves the
// younger stages' loads outstanding, one barrier per stage.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
template <int PER>
__device__ __forceinline__ void wait_stages_ahead(int ahead) {        // until at most `ahead` stages of PER loads each are outstanding
  static_assert(4 * PER < 64, "vmcnt is a 6-bit counter");
  switch (ahead) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<PER>(); break;
    case 2: wait_vmcnt<2 * PER>(); break;
    case 3: wait_vmcnt<3 * PER>(); break;
    default: wait_vmcnt<4 * PER>(); break;
  }
}

// PIPE (two buffers): the stage hand-over sits in front of the LAST k-step of a stage instead of behind it - the first fragments of
// stage t + 1 are read while the last MFMAs of stage t run.  Behind a barrier at the stage boundary all eight wavefronts read their
// first fragments at once (96 transpose reads = 48 KiB through the LDS pipe) with nothing on the matrix pipe: ~0.25 us per stage.
template <int F, int WI, int WJ, int LA, int LB, bool CS, bool PAIR = false, int KT = 64, int NBUF = 2, bool PIPE = false>
__global__ __launch_bounds__(64 * WI * WJ, WI * WJ == 8 ? 2 : 1) void gemm_tn_kernel(
    const uint16_t* __restrict__ A, long lda, const uint16_t* __restrict__ B, long ldb, float* __restrict__ ws, int M, int N1, int N2,
    int rows_per_split, int n_split, int dbg, const uint16_t* __restrict__ A2, const uint16_t* __restrict__ B2, long ldb2) {
  static_assert(!PAIR || (LA == 1 && LB == 0 && CS), "the pair: tile operand on the A side, row operand on the B side, column sums");
  constexpr int NW = WI * WJ, BM = 32 * F * WI, BN = 32 * F * WJ;
  constexpr int SUBA = BM / 32, SUBB = BN / 32, KS = KT / 16;         // KS: k-steps = 16-row DMA blocks per sub-image and stage
  constexpr uint32_t SUB = KT * 64;                                    // bytes of a sub-image (32 columns x KT rows)
  constexpr uint32_t A_BYTES = SUBA * SUB, B_BYTES = SUBB * SUB, STAGE = A_BYTES + B_BYTES;
  constexpr int NBLK = STAGE / 1024;
  static_assert(NBUF > 2 || NBLK % NW == 0, "DMA blocks per wavefront");
  static_assert(KT == 32 || KT == 64, "stage rows");
  constexpr int NDMA = (NBLK + NW - 1) / NW;                           // (ring: the last wavefronts may hold one block fewer)
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WJ, wj = wave - wi * WJ;
  const int TI = N1 / BM, TJ = N2 / BN;
  long lid;                                                            // logical workgroup id: consecutive ids share an XCD
  {
    const long L = blockIdx.x, G = gridDim.x;
    const long q = G / 8, r = G % 8, xcd = L % 8, k = L / 8;
    lid = xcd * q + (xcd < r ? xcd : r) + k;
  }
  constexpr int NPROB = PAIR ? 2 : 1;
  const int split = static_cast<int>(lid / (NPROB * TI * TJ));
  int tile = static_cast<int>(lid - static_cast<long>(split) * (NPROB * TI * TJ));
  const int prob = PAIR && tile >= TI * TJ ? 1 : 0;                    // (workgroup-uniform)
  if (PAIR && prob) { tile -= TI * TJ; A = A2; B = B2; ldb = ldb2; }
  const int ti = tile / TJ, tj = tile - ti * TJ;
  const int i0 = ti * BM, j0 = tj * BN;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int n_stage = (m_end - m_begin) / KT;

  const u32x4 rsa = make_rsrc4(A, static_cast<uint32_t>(LA == kRows ? static_cast<long>(M) * lda * 2 : static_cast<long>(M) * N1 * 2));
  const u32x4 rsb = make_rsrc4(B, static_cast<uint32_t>(LB == kRows ? static_cast<long>(M) * ldb * 2 : static_cast<long>(M) * N2 * 2));
  const uint32_t lds0 = lds_addr(lds);
  // lane part of a DMA source address.  ROWS: row lane / 4 of the block's 16, 16-byte piece lane % 4 of the sub-image's 64-byte row
  const uint32_t va = LA == kRows ? static_cast<uint32_t>((lane >> 2) * lda * 2 + (lane & 3) * 16) : static_cast<uint32_t>(lane * 16);
  const uint32_t vb = LB == kRows ? static_cast<uint32_t>((lane >> 2) * ldb * 2 + (lane & 3) * 16) : static_cast<uint32_t>(lane * 16);
  // part: the DMA instructions k with k % 4 == part (the loop below issues a quarter of a stage's DMA in front of each k-step's
  // MFMAs: a burst of all NDMA at the top of the stage holds the in-order wavefront at issue while the texture path drains); -1: all
  auto dma_stage = [&](int t, int buf, int part) {
    const uint32_t sb = lds0 + static_cast<uint32_t>(buf) * STAGE;
    const long m0 = m_begin + static_cast<long>(t) * KT;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      if (part >= 0 && (k % KS) != part) continue;
      const int b = wave + NW * k;                                     // wave-uniform block: (operand, sub-image, 16-row block)
      if (NBLK % NW != 0 && b >= NBLK) continue;
      if (b < SUBA * KS) {
        const int sub = b / KS, rb = b % KS;
        const long src = LA == kRows ? ((m0 + rb * 16) * lda + i0 + 32 * sub) * 2
                                     : (((m0 >> 5) + (rb >> 1)) * (N1 >> 5) + (i0 >> 5) + sub) * 2048 + (rb & 1) * 1024;
        dma_lds16(sb + static_cast<uint32_t>(sub) * SUB + static_cast<uint32_t>(rb * 1024), va, rsa, static_cast<uint32_t>(src));
      } else {
        const int bb = b - SUBA * KS, sub = bb / KS, rb = bb % KS;
        const long src = LB == kRows ? ((m0 + rb * 16) * ldb + j0 + 32 * sub) * 2
                                     : (((m0 >> 5) + (rb >> 1)) * (N2 >> 5) + (j0 >> 5) + sub) * 2048 + (rb & 1) * 1024;
        dma_lds16(sb + A_BYTES + static_cast<uint32_t>(sub) * SUB + static_cast<uint32_t>(rb * 1024), vb, rsb, static_cast<uint32_t>(src));
      }
    }
  };
  const uint32_t a_lane = lds0 + tn_lane_off<LA>(lane) + static_cast<uint32_t>(wi * F) * SUB;
  const uint32_t b_lane = lds0 + A_BYTES + tn_lane_off<LB>(lane) + static_cast<uint32_t>(wj * F) * SUB;

  f32x16 acc[F][F];
#pragma unroll
  for (int fi = 0; fi < F; ++fi)
#pragma unroll
    for (int fj = 0; fj < F; ++fj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fi][fj][r] = 0.f;
  // column sums: ONE more accumulator tile for all F blocks - block f is multiplied with a constant fragment whose column f (row f
  // for the sums of B) is ones, so its sums land in column (row) f of the tile
  f32x16 accs;
  bf16x8 ones[CS ? F : 1];
  // the column sums ride on the wavefronts of the first tile column (of A: x ones) / the first tile row (problem 1, of B: ones x)
  const bool cs_wave = CS && (prob == 0 ? (tj == 0 && wj == 0) : (ti == 0 && wi == 0));
  if constexpr (CS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[r] = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const uint32_t v = (lane & 31) == f ? 0x3f803f80u : 0u;
      const u32x4 t = {v, v, v, v};
      ones[f] = __builtin_bit_cast(bf16x8, t);
    }
  }

  // one stage's MFMAs from buffer `buf` (k-step ks + 1's fragments are read while the MFMAs of k-step ks run: two register sets);
  // `next`: issued in front of k-step ks - the share `ks` of a later stage's DMA
  auto multiply_stage = [&](int buf, auto&& next) {
    const uint32_t ab = a_lane + static_cast<uint32_t>(buf) * STAGE, bb = b_lane + static_cast<uint32_t>(buf) * STAGE;
    bf16x8 af[2][F], bfr[2][F];
    if (!(dbg & 2)) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        af[0][f] = tr_fragment(ab + static_cast<uint32_t>(f) * SUB, 64u);
        bfr[0][f] = tr_fragment(bb + static_cast<uint32_t>(f) * SUB, 64u);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      next(ks);
      if (dbg & 2) continue;
      if (ks + 1 < KS) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
          af[(ks + 1) & 1][f] = tr_fragment(ab + static_cast<uint32_t>(f) * SUB + static_cast<uint32_t>((ks + 1) * 1024), 64u);
          bfr[(ks + 1) & 1][f] = tr_fragment(bb + static_cast<uint32_t>(f) * SUB + static_cast<uint32_t>((ks + 1) * 1024), 64u);
        }
      }
#pragma unroll
      for (int fi = 0; fi < F; ++fi)
#pragma unroll
        for (int fj = 0; fj < F; ++fj)
          acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fi], bfr[ks & 1][fj], acc[fi][fj], 0, 0, 0);
      if constexpr (CS) {
        if (cs_wave) {
          if (PAIR && prob) {                                          // (a scalar branch: one of the two products, never both + a select)
#pragma unroll
            for (int fj = 0; fj < F; ++fj) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones[fj], bfr[ks & 1][fj], accs, 0, 0, 0);
          } else {
#pragma unroll
            for (int fi = 0; fi < F; ++fi) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fi], ones[fi], accs, 0, 0, 0);
          }
        }
      }
    }
  };
  if constexpr (NBUF == 2 && PIPE) {
    static_assert(KS % 2 == 0, "fragment register sets alternate across the stage boundary");
    bf16x8 af[2][F], bfr[2][F];
    auto read_frags = [&](int set, int buf, int ks) {
      const uint32_t ab = a_lane + static_cast<uint32_t>(buf) * STAGE + static_cast<uint32_t>(ks * 1024);
      const uint32_t bb = b_lane + static_cast<uint32_t>(buf) * STAGE + static_cast<uint32_t>(ks * 1024);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        af[set][f] = tr_fragment(ab + static_cast<uint32_t>(f) * SUB, 64u);
        bfr[set][f] = tr_fragment(bb + static_cast<uint32_t>(f) * SUB, 64u);
      }
    };
    if (n_stage > 0) dma_stage(0, 0, -1);
    wait_vmcnt<0>();
    __syncthreads();
    if (n_stage > 1) dma_stage(1, 1, 0);
    if (n_stage > 0) read_frags(0, 0, 0);
    for (int t = 0; t < n_stage; ++t) {
      const int buf = t & 1;
      const bool more = t + 1 < n_stage;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          if (more) dma_stage(t + 1, buf ^ 1, ks + 1);
          read_frags((ks + 1) & 1, buf, ks + 1);
        } else if (more) {
          // stage t + 1 has landed (its last share was issued a k-step ago); this wavefront's reads of stage t are complete, so behind
          // the barrier its buffer is free for stage t + 2
          wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __syncthreads();
          if (t + 2 < n_stage) dma_stage(t + 2, buf, 0);
          read_frags(0, buf ^ 1, 0);
        }
#pragma unroll
        for (int fi = 0; fi < F; ++fi)
#pragma unroll
          for (int fj = 0; fj < F; ++fj)
            acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fi], bfr[ks & 1][fj], acc[fi][fj], 0, 0, 0);
        if constexpr (CS) {
          if (cs_wave) {
            if (PAIR && prob) {
#pragma unroll
              for (int fj = 0; fj < F; ++fj) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones[fj], bfr[ks & 1][fj], accs, 0, 0, 0);
            } else {
#pragma unroll
              for (int fi = 0; fi < F; ++fi) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][fi], ones[fi], accs, 0, 0, 0);
            }
          }
        }
      }
    }
  } else if constexpr (NBUF == 2) {
    if (n_stage > 0) dma_stage(0, 0, -1);
    wait_vmcnt<0>();
    __syncthreads();
    for (int t = 0; t < n_stage; ++t) {
      const int buf = t & 1;
      const bool more = t + 1 < n_stage && !(dbg & 1);                 // (the other buffer's readers passed the barrier below)
      multiply_stage(buf, [&](int ks) { if (more) dma_stage(t + 1, buf ^ 1, ks); });
      wait_vmcnt<0>();
      __syncthreads();
    }
  } else {
    // ring of NBUF stages, D = NBUF - 1 of them in flight.  Iteration t: wait until stage t has landed (the D - 1 younger stages
    // stay outstanding), barrier (everybody's share of stage t is in LDS; everybody has finished stage t - 1, whose buffer is the
    // one of stage t + D), issue stage t + D over the k-steps of stage t.
    constexpr int D = NBUF - 1;
    const bool full = NBLK % NW == 0 || wave < NBLK % NW;              // this wavefront issues NDMA loads per stage (else NDMA - 1)
    for (int sg = 0; sg < D && sg < n_stage; ++sg) dma_stage(sg, sg, -1);
    int buf = 0;
    for (int t = 0; t < n_stage; ++t) {
      const int ahead = min(D - 1, n_stage - 1 - t);
      if (full) wait_stages_ahead<NDMA>(ahead); else wait_stages_ahead<(NDMA > 1 ? NDMA - 1 : 1)>(ahead);
      __syncthreads();
      const bool more = t + D < n_stage && !(dbg & 1);
      const int nbuf = buf == 0 ? NBUF - 1 : buf - 1;                  // (t + D) % NBUF
      multiply_stage(buf, [&](int ks) { if (more) dma_stage(t + D, nbuf, ks); });
      buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
  }
  // partial tile -> ws[split][N1][N2] (then [N1] column sums of A); the pair: [D0][N1 sums][T][N2 sums] per split
  const long len_d = static_cast<long>(N1) * N2;
  float* out = ws + static_cast<long>(split) * (PAIR ? 2 * len_d + N1 + N2 : len_d + (CS ? N1 : 0)) + (PAIR && prob ? len_d + N1 : 0);
  const int col = lane & 31, kb = lane >> 5;
#pragma unroll
  for (int fi = 0; fi < F; ++fi)
#pragma unroll
    for (int fj = 0; fj < F; ++fj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 32 * (wi * F + fi) + (r & 3) + 8 * (r >> 2) + 4 * kb;
        const int j = j0 + 32 * (wj * F + fj) + col;
        out[static_cast<long>(i) * N2 + j] = acc[fi][fj][r];
      }
  if constexpr (CS) {
    if (PAIR && prob) {
      // ones x B: row f of the tile = the column sums of B's block f (accumulator value f of the lanes of k-half 0)
      if (cs_wave && kb == 0) {
#pragma unroll
        for (int fj = 0; fj < F; ++fj) out[len_d + j0 + 32 * (wj * F + fj) + col] = accs[fj];
      }
    } else if (cs_wave && col < F) {
#pragma unroll
      for (int r = 0; r < 16; ++r) out[len_d + i0 + 32 * (wi * F + col) + (r & 3) + 8 * (r >> 2) + 4 * kb] = accs[r];
    }
  }
}

// The pair's fixed-order sum: D0 [N1][N2] and its [N1] sums as they lie in the partials, T [N1][N2] transposed into D1 [N2][N1]
// through LDS (32 x 32 tiles: 128-byte segments on both sides), then its [N2] sums.  1024 threads = 4 split-lanes x 256 positions
// (a float4 each): lane q adds the splits q, q + 4, ... in order, the four lanes are added as (0 + 1) + (2 + 3).
__global__ __launch_bounds__(1024) void gemm_tn_pair_reduce_kernel(const float* __restrict__ ws, float* __restrict__ D0, float* __restrict__ cs0,
                                                                   float* __restrict__ D1, float* __restrict__ cs1, int N1, int N2, int n_split) {
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  __shared__ f32x4 part[4][256];
  __shared__ float tr[32][33];
  const int q = threadIdx.x >> 8, pos = threadIdx.x & 255;
  const long len_d = static_cast<long>(N1) * N2, part_len = 2 * len_d + N1 + N2;
  const int tiles_j = N2 / 32, tiles = (N1 / 32) * tiles_j;
  int b = blockIdx.x;
  const bool mat = b < 2 * tiles;
  const int prob = mat && b >= tiles ? 1 : 0;
  long off, e = 0;                                                     // this thread's float4 inside a split's partials; -1: none
  int ti = 0, tj = 0;
  if (mat) {
    if (prob) b -= tiles;
    ti = b / tiles_j, tj = b - ti * tiles_j;
    off = (prob ? len_d + N1 : 0) + static_cast<long>(ti * 32 + (pos >> 3)) * N2 + tj * 32 + (pos & 7) * 4;
  } else {
    e = (static_cast<long>(b - 2 * tiles) * 256 + pos) * 4;            // the N1 + N2 sums: [N1] behind D0, [N2] behind T
    off = e < N1 ? len_d + e : e < N1 + N2 ? 2 * len_d + e : -1;
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (off >= 0) {
    const float* p = ws + off;
    int k = q;
    for (; k + 12 < n_split; k += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + k * part_len), bq = *reinterpret_cast<const f32x4*>(p + (k + 4) * part_len),
                  c = *reinterpret_cast<const f32x4*>(p + (k + 8) * part_len), d = *reinterpret_cast<const f32x4*>(p + (k + 12) * part_len);
      s += a; s += bq; s += c; s += d;
    }
    for (; k < n_split; k += 4) s += *reinterpret_cast<const f32x4*>(p + k * part_len);
  }
  part[q][pos] = s;
  __syncthreads();
  if (q == 0) s = (part[0][pos] + part[1][pos]) + (part[2][pos] + part[3][pos]);
  if (!mat) {
    if (q == 0 && off >= 0) {
      if (e < N1) *reinterpret_cast<f32x4*>(cs0 + e) = s;
      else *reinterpret_cast<f32x4*>(cs1 + (e - N1)) = s;
    }
    return;
  }
  if (!prob) {
    if (q == 0) *reinterpret_cast<f32x4*>(D0 + off) = s;
    return;
  }
  if (q == 0) {
    const int il = pos >> 3, jl = (pos & 7) * 4;
    tr[il][jl] = s.x; tr[il][jl + 1] = s.y; tr[il][jl + 2] = s.z; tr[il][jl + 3] = s.w;
  }
  __syncthreads();
  if (q == 0) {
    const int jl = pos >> 3, il = (pos & 7) * 4;
    const f32x4 v = {tr[il][jl], tr[il + 1][jl], tr[il + 2][jl], tr[il + 3][jl]};
    *reinterpret_cast<f32x4*>(D1 + static_cast<long>(tj * 32 + jl) * N1 + ti * 32 + il) = v;
  }
}

// D (then the optional column-sum vector) = sum of the n_split partial results, in split order (deterministic); four consecutive
// outputs per thread.  part_len = floats per split = len_d + len_cs.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ D, float* __restrict__ cs,
                                                             long len_d, long part_len, int n_split) {
  const long v = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (v * 4 >= part_len) return;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  const f32x4* p = reinterpret_cast<const f32x4*>(ws) + v;
  const long stride = part_len / 4;
  f32x4 s = p[0];
  int k = 1;
  for (; k + 3 < n_split; k += 4) {
    const f32x4 a = p[static_cast<long>(k) * stride], b = p[static_cast<long>(k + 1) * stride], c = p[static_cast<long>(k + 2) * stride],
                d = p[static_cast<long>(k + 3) * stride];
    s += a; s += b; s += c; s += d;
  }
  for (; k < n_split; ++k) s += p[static_cast<long>(k) * stride];
  if (v * 4 < len_d) reinterpret_cast<f32x4*>(D)[v] = s;
  else reinterpret_cast<f32x4*>(cs)[v - len_d / 4] = s;
}

template <int F, int WI, int WJ, int LA, int LB, bool CS>
int launch_gemm_tn(const uint16_t* A, long lda, const uint16_t* B, long ldb, float* D, float* cs, float* ws, int M, int N1, int N2,
                   int rows_per_split, int n_split, hipStream_t s) {
  constexpr int BM = 32 * F * WI, BN = 32 * F * WJ;
  constexpr size_t lds_bytes = 2 * (BM / 32 + BN / 32) * 64 * 64;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<F, WI, WJ, LA, LB, CS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    return true;
  }();
  (void)attr;
  const long grid = static_cast<long>(N1 / BM) * (N2 / BN) * n_split;
  // timing experiments (1 = no DMA after stage 0, 2 = no MFMAs, 4 = no reduce: all return WRONG gradients) exist in measurement
  // builds only (`make EXTRA=-DTN_ABLATE=1`): a stray environment variable must not be able to corrupt training
#if TN_ABLATE
  static const int dbg = getenv("APGD_TN_DBG") ? atoi(getenv("APGD_TN_DBG")) : 0;
#else
  constexpr int dbg = 0;
#endif
  // (a single split with no column sums writes D directly; everything else goes through the workspace and the fixed-order sum)
  const bool direct = n_split == 1 && !CS;
  hipLaunchKernelGGL((gemm_tn_kernel<F, WI, WJ, LA, LB, CS>), dim3(static_cast<unsigned>(grid)), dim3(64 * WI * WJ), lds_bytes, s, A, lda, B,
                     ldb, direct ? D : ws, M, N1, N2, rows_per_split, n_split, dbg, static_cast<const uint16_t*>(nullptr),
                     static_cast<const uint16_t*>(nullptr), 0L);
  int rc = launch_status();
  if (rc || direct || (dbg & 4)) return rc;
  const long len_d = static_cast<long>(N1) * N2, part = len_d + (CS ? N1 : 0);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(static_cast<unsigned>((part / 4 + 255) / 256)), dim3(256), 0, s, ws, D, cs, len_d, part, n_split);
  return launch_status();
}

// cnx_runtime_switch(CNX_SWITCH_TN_PAIR_RING) - the stage loop of the pair (all bit-identical; tools/gemm_tn_bench.py, us per block at batch 256):
//   0 = two 64-row buffers, hand-over at the stage boundary      C = 96: 283   128: 106   192: 175   256: 88   384: 148
//   2 = ring of 32-row stages, three or more in flight                   275        103        179        96        173
//   1, 3 = two buffers, hand-over one k-step early (default)             269        105        173        83        148
// Deeper prefetch and the hidden first-fragment reads return 5 % on the HBM-bound shapes and nothing at C = 384: there the kernel sits
// at what a CU ingests (28.9 GB/s per CU = 7.4 TB/s over the chip, 52 % of it L2 hits), as every GEMM-shaped kernel of this library and
// hipBLASLt's own (MT256x256x64: 990 TFLOP/s at 128 FLOP per staged byte = 7.7 TB/s) - FLOP per staged byte is the only lever left.
int g_pair_ring = 1;

template <int F, int WI, int WJ, int KT, int NBUF, bool PIPE = false>
int launch_gemm_tn_pair(const uint16_t* A0, const uint16_t* B0, long ldb0, const uint16_t* A1, const uint16_t* B1, long ldb1, float* D0,
                        float* cs0, float* D1, float* cs1, float* ws, int M, int N1, int N2, int rows_per_split, int n_split, hipStream_t s) {
  constexpr int BM = 32 * F * WI, BN = 32 * F * WJ;
  constexpr size_t lds_bytes = static_cast<size_t>(NBUF) * (BM / 32 + BN / 32) * KT * 64;
  static_assert(lds_bytes <= 160 * 1024, "LDS");
  auto kfn = gemm_tn_kernel<F, WI, WJ, kAcc, kRows, true, true, KT, NBUF, PIPE>;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<F, WI, WJ, kAcc, kRows, true, true, KT, NBUF, PIPE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  const long grid = 2L * (N1 / BM) * (N2 / BN) * n_split;
  hipLaunchKernelGGL(kfn, dim3(static_cast<unsigned>(grid)), dim3(64 * WI * WJ), lds_bytes, s,
                     A0, static_cast<long>(N1), B0, ldb0, ws, M, N1, N2, rows_per_split, n_split, 0, A1, B1, ldb1);
  int rc = launch_status();
  if (rc) return rc;
  const long blocks = 2L * (N1 / 32) * (N2 / 32) + (N1 + N2 + 1023) / 1024;
  hipLaunchKernelGGL(gemm_tn_pair_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(1024), 0, s, ws, D0, cs0, D1, cs1, N1, N2, n_split);
  return launch_status();
}

// tile shape for (N1, N2): 0 = none.  code = F * 100 + WI * 10 + WJ
inline int gemm_tn_shape(int N1, int N2) {
  const int cand[8][3] = {{3, 4, 2}, {3, 2, 4}, {2, 4, 2}, {2, 2, 4}, {3, 4, 1}, {3, 1, 4}, {2, 4, 1}, {2, 1, 4}};
  for (const auto& c : cand)
    if (N1 % (32 * c[0] * c[1]) == 0 && N2 % (32 * c[0] * c[2]) == 0) return c[0] * 100 + c[1] * 10 + c[2];
  return 0;
}
// split of M over workgroups: about one workgroup per CU (a workgroup fills a CU's LDS), whole stages of 64 rows, at least 4 stages
// per split.  Measured (tools/gemm_tn_bench.py): 512 workgroups double the partial results' traffic and run 20 - 30 % longer.
inline void gemm_tn_split(int M, int tiles, int* rows_per_split, int* n_split) {
  const int stages = M / 64;
#if TN_ABLATE
  static const int wgs = getenv("APGD_TN_WGS") ? atoi(getenv("APGD_TN_WGS")) : 256;   // (measurement builds: the split sweep)
#else
  constexpr int wgs = 256;
#endif
  int want = (wgs + tiles - 1) / tiles;
  if (want < 1) want = 1;
  int per = (stages + want - 1) / want;
  if (per < 4) per = stages < 4 ? stages : 4;
  *rows_per_split = per * 64;
  *n_split = (stages + per - 1) / per;
}

// out[j] = sum over parts (fixed order: 8 interleaved running sums, then a tree) of ws[part * len + j]; one thread per j and
// part-lane, 32 part-lanes per output.  map: j -> destination index (the caller's layouts differ from the partials'), or identity.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                           int len, int split, int nparts, int tr_rows, int tr_cols) {
  __shared__ float part[32][9];
  const int jl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + jl;
  float s = 0.f;
  if (j < len)
    for (int p = pl; p < nparts; p += 32) s += ws[static_cast<long>(p) * len + j];
  part[pl][jl] = s;
  __syncthreads();
  if (pl == 0 && j < len) {
    float t[32];
#pragma unroll
    for (int g = 0; g < 32; ++g) t[g] = part[g][jl];
#pragma unroll
    for (int w = 16; w > 0; w >>= 1)
#pragma unroll
      for (int g = 0; g < w; ++g) t[g] += t[g + w];
    if (j < split) {
      // the first `split` values are a [tr_rows][tr_cols] matrix that the caller wants transposed
      const int r = j / tr_cols, cc = j - r * tr_cols;
      out0[tr_rows > 0 ? cc * tr_rows + r : j] = t[0];
    } else if (out1) {
      out1[j - split] = t[0];
    }
  }
}

}  // namespace

int tn_pair_ring_switch(int value) {
  const int prev = g_pair_ring;
  if (value >= 0) g_pair_ring = value > 3 ? 1 : value;
  return prev;
}

extern "C" {

int64_t cnx_stem_conv_wgrad_ws_floats(int32_t P) { return 1024L * 28 * P; }

int cnx_stem_conv_wgrad(const float* x, const void* dy, float* dw, float* dbias, float* ws, int64_t N, int32_t H, int32_t W, int32_t P,
                        void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if (!(P == 48 || P == 64 || P == 96)) return APGD_ERR_ARG;
  if ((H & 1) || (W & 1) || static_cast<long>(N) * 3 * H * W >= (1L << 30)) return APGD_ERR_ARG;
  if (!dw || !ws || (N > 0 && (!x || !dy))) return APGD_ERR_NULL;
  const int OH = H / 2, OW = W / 2;
  const long total = N * OH * OW;
  if (total * P * 2 >= (1L << 32)) return APGD_ERR_ARG;                // 32-bit byte offsets into dy
  hipStream_t s = as_stream(stream);
  long nchunks = (total + 15) / 16;
  long nwg = (nchunks + 4 * 6 - 1) / (4 * 6);                          // ~6 chunks per wavefront, at most 1024 partial results
  if (nwg > 1024) nwg = 1024;
  if (nwg < 1) nwg = 1;
  const dim3 grid(static_cast<unsigned>(nwg)), block(256);
  const auto* d = static_cast<const uint16_t*>(dy);
  const bool row16 = OW % 16 == 0;
#define STEM_WG(PP)                                                                                                          \
  if (row16) hipLaunchKernelGGL((stem_conv_wgrad_kernel<PP, true>), grid, block, 0, s, x, d, ws, total, static_cast<int>(N), H, W, OH, OW);  \
  else hipLaunchKernelGGL((stem_conv_wgrad_kernel<PP, false>), grid, block, 0, s, x, d, ws, total, static_cast<int>(N), H, W, OH, OW);
  if (P == 48) { STEM_WG(48) } else if (P == 64) { STEM_WG(64) } else { STEM_WG(96) }
#undef STEM_WG
  int rc = launch_status();
  if (rc) return rc;
  // partial layout [28][P]: rows 0..26 = (ci, kh, kw) x P  ->  dw[p][27] ([P, 3, 3, 3]); row 27 -> dbias[P]
  const int len = 28 * P;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((len + 7) / 8), dim3(256), 0, s, ws, dw, dbias, len, 27 * P, static_cast<int>(nwg), 27, P);
  return launch_status();
}

int cnx_conv3x3s2_wgrad_supported(int32_t CI, int32_t CO, int32_t H, int32_t W) {
  if (!(CO == 96 && (CI == 48 || CI == 64)) || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
  const int KS = (W / 2 + 15) / 16, WP = 32 * KS + 3;
  return ((3L * WP * CI + 16L * KS * CO) * 4 <= 150 * 1024 && (W * CI) % 8 == 0) ? 1 : 0;   // two buffers of both row images in LDS
}

int64_t cnx_conv3x3s2_wgrad_ws_floats(int32_t CI, int32_t CO) { return 512L * (9L * CI * CO + CO); }

int cnx_conv3x3s2_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws, int64_t N, int32_t H, int32_t W, int32_t CI,
                        int32_t CO, void* stream) {
  if (N < 0) return APGD_ERR_SIZE;
  if (!cnx_conv3x3s2_wgrad_supported(CI, CO, H, W)) return APGD_ERR_ARG;
  if (!dw || !ws || (N > 0 && (!x || !dy))) return APGD_ERR_NULL;
  if (static_cast<long>(N) * H * W * CI * 2 >= (1L << 32)) return APGD_ERR_ARG;     // 32-bit byte offsets
  hipStream_t s = as_stream(stream);
  const int KS = (W / 2 + 15) / 16, WP = 32 * KS + 3;
  const size_t lds_bytes = (3L * WP * CI + 16L * KS * CO) * 4;
  const long rows = N * (H / 2);
  long nwg = rows < 256 ? rows : 256;                                  // one resident workgroup (eight wavefronts) per CU
  if (nwg < 1) nwg = 1;
  const auto* xp = static_cast<const uint16_t*>(x);
  const auto* dp = static_cast<const uint16_t*>(dy);
  static const bool lds_attr = [] {                                    // dynamic LDS beyond 64 KB needs the attribute (once per process)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wgrad_kernel<48, 96, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wgrad_kernel<64, 96, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)lds_attr;
  if (CI == 48)
    hipLaunchKernelGGL((conv2_wgrad_kernel<48, 96, 2>), dim3(static_cast<unsigned>(nwg)), dim3(kW2 * 64), lds_bytes, s, xp, dp, ws, static_cast<int>(N), H, W);
  else
    hipLaunchKernelGGL((conv2_wgrad_kernel<64, 96, 3>), dim3(static_cast<unsigned>(nwg)), dim3(kW2 * 64), lds_bytes, s, xp, dp, ws, static_cast<int>(N), H, W);
  int rc = launch_status();
  if (rc) return rc;
  // partials [CO][9 CI] (+ [CO]) -> dw in the weight's channels-last order [CO][kh][kw][CI], dbias[CO]
  const int len = CO * 9 * CI + CO;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((len + 7) / 8), dim3(256), 0, s, ws, dw, dbias, len, CO * 9 * CI, static_cast<int>(nwg), 0, 1);
  return launch_status();
}

int cnx_gemm_tn_supported(int64_t M, int32_t N1, int32_t N2) {
  return (M > 0 && M % 64 == 0 && M < (1L << 31) && gemm_tn_shape(N1, N2) != 0) ? 1 : 0;
}

int64_t cnx_gemm_tn_ws_floats(int64_t M, int32_t N1, int32_t N2) {
  const int code = gemm_tn_shape(N1, N2);
  if (!code || M <= 0 || M % 64 != 0) return 0;
  const int F = code / 100, WI = (code / 10) % 10, WJ = code % 10;
  int rows, ns;
  gemm_tn_split(static_cast<int>(M), (N1 / (32 * F * WI)) * (N2 / (32 * F * WJ)), &rows, &ns);
  return static_cast<int64_t>(ns) * (static_cast<int64_t>(N1) * N2 + N1);
}

int cnx_gemm_tn_ex(const void* A, int64_t lda, int32_t a_layout, const void* B, int64_t ldb, int32_t b_layout, float* D, float* colsum_a,
                   float* ws, int64_t M, int32_t N1, int32_t N2, void* stream) {
  if (M < 0 || N1 <= 0 || N2 <= 0) return APGD_ERR_SIZE;
  if (!A || !B || !D || !ws) return APGD_ERR_NULL;
  if (!cnx_gemm_tn_supported(M, N1, N2)) return APGD_ERR_ARG;
  if ((a_layout != CNX_TN_ROWS && a_layout != CNX_TN_ACC) || (b_layout != CNX_TN_ROWS && b_layout != CNX_TN_ACC)) return APGD_ERR_ARG;
  if (a_layout == CNX_TN_ACC && b_layout == CNX_TN_ACC) return APGD_ERR_ARG;            // (no caller: not instantiated)
  if (a_layout == CNX_TN_ROWS && (lda < N1 || lda % 8 != 0)) return APGD_ERR_ARG;
  if (b_layout == CNX_TN_ROWS && (ldb < N2 || ldb % 8 != 0)) return APGD_ERR_ARG;
  if (colsum_a && a_layout == CNX_TN_ROWS && b_layout == CNX_TN_ROWS) return APGD_ERR_ARG;   // column sums come with an ACC operand pair
  if (!colsum_a && (a_layout == CNX_TN_ACC || b_layout == CNX_TN_ACC)) return APGD_ERR_NULL;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(ws) |
       reinterpret_cast<uintptr_t>(colsum_a)) % 16 != 0)
    return APGD_ERR_ARG;
  const int64_t la = a_layout == CNX_TN_ROWS ? lda : N1, lb = b_layout == CNX_TN_ROWS ? ldb : N2;
  if (M * la * 2 >= (1L << 32) || M * lb * 2 >= (1L << 32)) return APGD_ERR_ARG;        // 32-bit byte offsets
  const int code = gemm_tn_shape(N1, N2);
  const int F = code / 100, WI = (code / 10) % 10, WJ = code % 10;
  int rows, ns;
  gemm_tn_split(static_cast<int>(M), (N1 / (32 * F * WI)) * (N2 / (32 * F * WJ)), &rows, &ns);
  hipStream_t s = as_stream(stream);
  const auto* a = static_cast<const uint16_t*>(A);
  const auto* b = static_cast<const uint16_t*>(B);
  const int m = static_cast<int>(M);
#define TN_CASE(FF, II, JJ)                                                                                                        \
  if (code == FF * 100 + II * 10 + JJ) {                                                                                           \
    if (a_layout == CNX_TN_ACC) return launch_gemm_tn<FF, II, JJ, kAcc, kRows, true>(a, lda, b, ldb, D, colsum_a, ws, m, N1, N2, rows, ns, s); \
    if (b_layout == CNX_TN_ACC) return launch_gemm_tn<FF, II, JJ, kRows, kAcc, true>(a, lda, b, ldb, D, colsum_a, ws, m, N1, N2, rows, ns, s); \
    return launch_gemm_tn<FF, II, JJ, kRows, kRows, false>(a, lda, b, ldb, D, nullptr, ws, m, N1, N2, rows, ns, s);                 \
  }
  TN_CASE(3, 4, 2) TN_CASE(3, 2, 4) TN_CASE(2, 4, 2) TN_CASE(2, 2, 4) TN_CASE(3, 4, 1) TN_CASE(3, 1, 4) TN_CASE(2, 4, 1) TN_CASE(2, 1, 4)
#undef TN_CASE
  return APGD_ERR_ARG;
}

// the pair runs on the tiles with WI >= WJ (what gemm_tn_shape picks for N1 = 4 N2, the two weight gradients of a block)
static int gemm_tn_pair_shape(int N1, int N2) {
  const int code = gemm_tn_shape(N1, N2);
  return (code / 10) % 10 >= code % 10 ? code : 0;
}

int cnx_gemm_tn_pair_supported(int64_t M, int32_t N1, int32_t N2) {
  return (M > 0 && M % 64 == 0 && M < (1L << 31) && N1 > 0 && N2 > 0 && gemm_tn_pair_shape(N1, N2) != 0) ? 1 : 0;
}

int64_t cnx_gemm_tn_pair_ws_floats(int64_t M, int32_t N1, int32_t N2) {
  if (!cnx_gemm_tn_pair_supported(M, N1, N2)) return 0;
  const int code = gemm_tn_pair_shape(N1, N2);
  const int F = code / 100, WI = (code / 10) % 10, WJ = code % 10;
  int rows, ns;
  gemm_tn_split(static_cast<int>(M), 2 * (N1 / (32 * F * WI)) * (N2 / (32 * F * WJ)), &rows, &ns);
  return static_cast<int64_t>(ns) * (2 * static_cast<int64_t>(N1) * N2 + N1 + N2);
}

int cnx_gemm_tn_pair(const void* A0, const void* B0, int64_t ldb0, const void* A1, const void* B1, int64_t ldb1, float* D0, float* colsum_a0,
                     float* D1, float* colsum_b1, float* ws, int64_t M, int32_t N1, int32_t N2, void* stream) {
  if (M < 0 || N1 <= 0 || N2 <= 0) return APGD_ERR_SIZE;
  if (!A0 || !B0 || !A1 || !B1 || !D0 || !D1 || !colsum_a0 || !colsum_b1 || !ws) return APGD_ERR_NULL;
  if (!cnx_gemm_tn_pair_supported(M, N1, N2)) return APGD_ERR_ARG;
  if (ldb0 < N2 || ldb0 % 8 != 0 || ldb1 < N2 || ldb1 % 8 != 0) return APGD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(A0) | reinterpret_cast<uintptr_t>(B0) | reinterpret_cast<uintptr_t>(A1) | reinterpret_cast<uintptr_t>(B1) |
       reinterpret_cast<uintptr_t>(D0) | reinterpret_cast<uintptr_t>(D1) | reinterpret_cast<uintptr_t>(colsum_a0) |
       reinterpret_cast<uintptr_t>(colsum_b1) | reinterpret_cast<uintptr_t>(ws)) % 16 != 0)
    return APGD_ERR_ARG;
  if (M * N1 * 2 >= (1L << 32) || M * ldb0 * 2 >= (1L << 32) || M * ldb1 * 2 >= (1L << 32)) return APGD_ERR_ARG;   // 32-bit byte offsets
  const int code = gemm_tn_pair_shape(N1, N2);
  const int F = code / 100, WI = (code / 10) % 10, WJ = code % 10;
  int rows, ns;
  gemm_tn_split(static_cast<int>(M), 2 * (N1 / (32 * F * WI)) * (N2 / (32 * F * WJ)), &rows, &ns);
  hipStream_t s = as_stream(stream);
  const auto* a0 = static_cast<const uint16_t*>(A0);
  const auto* b0 = static_cast<const uint16_t*>(B0);
  const auto* a1 = static_cast<const uint16_t*>(A1);
  const auto* b1 = static_cast<const uint16_t*>(B1);
  const int m = static_cast<int>(M);
  // (stage rows, stages in LDS) of the ring per tile: 36 / 24 / 30 / 20 KiB per 32-row stage -> 144 / 144 / 150 / 120 KiB
#define TN_PAIR(FF, II, JJ, NB)                                                                                                       \
  if (code == FF * 100 + II * 10 + JJ) {                                                                                              \
    if (g_pair_ring == 2) return launch_gemm_tn_pair<FF, II, JJ, 32, NB>(a0, b0, ldb0, a1, b1, ldb1, D0, colsum_a0, D1, colsum_b1, ws, m, N1, N2, rows, ns, s); \
    if (g_pair_ring & 1) return launch_gemm_tn_pair<FF, II, JJ, 64, 2, true>(a0, b0, ldb0, a1, b1, ldb1, D0, colsum_a0, D1, colsum_b1, ws, m, N1, N2, rows, ns, s); \
    return launch_gemm_tn_pair<FF, II, JJ, 64, 2>(a0, b0, ldb0, a1, b1, ldb1, D0, colsum_a0, D1, colsum_b1, ws, m, N1, N2, rows, ns, s);     \
  }
  TN_PAIR(3, 4, 2, 4) TN_PAIR(2, 4, 2, 6) TN_PAIR(3, 4, 1, 5) TN_PAIR(2, 4, 1, 6)
#undef TN_PAIR
  return APGD_ERR_ARG;
}

int cnx_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* D, float* ws, int64_t M, int32_t N1, int32_t N2, void* stream) {
  return cnx_gemm_tn_ex(A, lda, CNX_TN_ROWS, B, ldb, CNX_TN_ROWS, D, nullptr, ws, M, N1, N2, stream);
}

}  // extern "C"
