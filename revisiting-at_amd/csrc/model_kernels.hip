// model_kernels.hip — gfx950 kernels for the ConvNeXt block / ConvStem pieces of the hot path
// (C ABI: include/convnext_hip.h).  Reference arithmetic: /root/reference/models/convnext.py:28-41
// (depthwise 7x7 + LayerNorm) and /root/reference/utils_architecture.py:76-81 (+GELU in the stems).
//
// All of these are bandwidth-bound (no contraction over channels): channels-last rows, 16-byte
// per-lane accesses coalesced along C, fp32 accumulation, wavefront-64 shuffles + LDS for the
// per-row statistics, deterministic two-stage reductions for parameter gradients.  No MFMA here.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "apgd_hip.h"
#include "convnext_hip.h"
#include "dw_internal.h"

// timing experiments (APGD_DW_DBG) are compiled in only with -DDW_ABLATE=1: a run-time flag test inside a stencil loop is a
// branch per use (see profiles/r02_fused_mlp_study.md)
#ifndef DW_ABLATE
#define DW_ABLATE 0
#endif
#define DW_DBG(dbg, bit) (DW_ABLATE && ((dbg) & (bit)))

namespace {

constexpr int kWave = 64;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

// 4 consecutive channels as fp32, from fp32 or bf16 storage
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const uint16_t* p) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                     __uint_as_float(t.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(uint16_t* p, float4 v) {
  uint2 t;
  t.x = static_cast<uint32_t>(f2bf(v.x)) | (static_cast<uint32_t>(f2bf(v.y)) << 16);
  t.y = static_cast<uint32_t>(f2bf(v.z)) | (static_cast<uint32_t>(f2bf(v.w)) << 16);
  *reinterpret_cast<uint2*>(p) = t;
}
__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& x) {
  a.x = fmaf(w.x, x.x, a.x); a.y = fmaf(w.y, x.y, a.y); a.z = fmaf(w.z, x.z, a.z); a.w = fmaf(w.w, x.w, a.w);
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 (forward, and input-gradient via the rotated filter)
//   block = (LC lanes over channels [4 each], PY strips); a strip = RH output rows x TW output columns.
//   Filter chunk [49][CC] lives in LDS (rotated at load time when flip).  Each input row segment is
//   loaded once into registers and feeds up to RH output rows x 7 taps (sliding window along W).
// ------------------------------------------------------------------------------------------------
constexpr int kTW = 7;    // 56, 28, 14, 7 (ConvNeXt @224) are multiples of 7; other widths use the tail predicate
constexpr int kRH = 2;
constexpr int kCC = 192;  // channels per workgroup (<= 48 lanes x 4)

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dwconv7x7_kernel(const TI* __restrict__ x, const float* __restrict__ w49c,
                                                        const float* __restrict__ bias, const float* __restrict__ add,
                                                        TO* __restrict__ out, int H, int W, int C, int flip,
                                                        int tiles_h, int tiles_w, long n_strips) {
  // a call with any bf16 operand is the autocast convolution: input and filter rounded to bf16, fp32 accumulation
  // (same numerics as the packed-dot kernel the ConvNeXt shapes take)
  constexpr bool kMixed = sizeof(TI) == 2 || sizeof(TO) == 2;
  extern __shared__ float wl[];                       // [49][cc]
  const int cbase = blockIdx.y * kCC;
  const int cc = min(kCC, C - cbase);
  const int lc = threadIdx.x, py = threadIdx.y;
  const int nthr = blockDim.x * blockDim.y, tid = py * blockDim.x + lc;
  for (int i = tid; i < 49 * cc; i += nthr) {
    const int tap = i / cc, c = i - tap * cc;
    float wv = w49c[tap * C + cbase + c];
    if constexpr (kMixed) wv = __uint_as_float(static_cast<uint32_t>(f2bf(wv)) << 16);
    wl[(flip ? 48 - tap : tap) * cc + c] = wv;
  }
  __syncthreads();
  const long s = static_cast<long>(blockIdx.x) * blockDim.y + py;
  const int cl = lc * 4;
  if (s >= n_strips || cl >= cc) return;
  const int c0 = cbase + cl;
  const int th = static_cast<int>(s % tiles_h);
  const long r1 = s / tiles_h;
  const int tw = static_cast<int>(r1 % tiles_w);
  const long n = r1 / tiles_w;
  const int h0 = th * kRH, w0 = tw * kTW;

  float4 acc[kRH][kTW];
  const float4 b4 = bias ? load4(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int oh = 0; oh < kRH; ++oh)
#pragma unroll
    for (int t = 0; t < kTW; ++t) acc[oh][t] = b4;

  const TI* xn = x + (n * H) * static_cast<long>(W) * C + c0;
#pragma unroll
  for (int r = 0; r < kRH + 6; ++r) {
    const int hh = h0 - 3 + r;
    if (hh < 0 || hh >= H) continue;
    const TI* xr = xn + static_cast<long>(hh) * W * C;
    float4 in[kTW + 6];
#pragma unroll
    for (int j = 0; j < kTW + 6; ++j) {
      const int col = w0 - 3 + j;
      in[j] = (col >= 0 && col < W) ? load4(xr + static_cast<long>(col) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (kMixed && sizeof(TI) == 4) {
        in[j].x = __uint_as_float(static_cast<uint32_t>(f2bf(in[j].x)) << 16); in[j].y = __uint_as_float(static_cast<uint32_t>(f2bf(in[j].y)) << 16);
        in[j].z = __uint_as_float(static_cast<uint32_t>(f2bf(in[j].z)) << 16); in[j].w = __uint_as_float(static_cast<uint32_t>(f2bf(in[j].w)) << 16);
      }
    }
#pragma unroll
    for (int oh = 0; oh < kRH; ++oh) {
      const int kh = r - oh;
      if (kh < 0 || kh > 6) continue;
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const float4 wv = *reinterpret_cast<const float4*>(&wl[(kh * 7 + kw) * cc + cl]);
#pragma unroll
        for (int t = 0; t < kTW; ++t) fma4(acc[oh][t], wv, in[t + kw]);
      }
    }
  }
#pragma unroll
  for (int oh = 0; oh < kRH; ++oh) {
    const int h = h0 + oh;
    if (h >= H) continue;
#pragma unroll
    for (int t = 0; t < kTW; ++t) {
      const int w = w0 + t;
      if (w >= W) continue;
      const long off = ((n * H + h) * static_cast<long>(W) + w) * C + c0;
      float4 v = acc[oh][t];
      if (add) { const float4 a = load4(add + off); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      store4(out + off, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7, LDS-tiled variant (the one the ConvNeXt shapes take).
//   The strip kernel above reads every input element ~7x through the vector L1 (13 x 8 loads per 14 outputs): it is
//   bound by load-instruction issue, not by HBM.  Here a workgroup stages a zero-padded (TH+6) x (W+6) x CC input
//   tile in LDS once (each global element is read once per workgroup, 16 bytes per lane, coalesced along C), then
//   every thread computes a 2 x 7 output strip for 4 channels out of LDS with the same sliding register window.
//   Staging type TS: bf16 when the input or the output is bf16 (autocast rounds the convolution's input to bf16
//   anyway), fp32 for the all-fp32 instantiation.  Lane order: channel group fastest, then strips along W, so the
//   4 (or 2) strips of a 32-lane group hit distinct LDS bank quarters (7 positions x CC x 2 B = 192 mod 256).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_put4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void lds_put4(uint16_t* p, float4 v) { store4(p, v); }

template <typename TI, typename TO, typename TS, int CC>
__global__ __launch_bounds__(512) void dwconv7x7_tile_kernel(const TI* __restrict__ x, const float* __restrict__ w49c,
                                                              const float* __restrict__ bias, const float* __restrict__ add,
                                                              TO* __restrict__ out, int H, int W, int C, int flip, int TH,
                                                              int tiles_h) {
  constexpr int LC = CC / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                              // [49][CC]
  TS* tile = reinterpret_cast<TS*>(smem + 49 * CC * sizeof(float));        // [TH+6][W+6][CC]
  const int Wp = W + 6;
  const long n = blockIdx.x / tiles_h;
  const int h0 = static_cast<int>(blockIdx.x % tiles_h) * TH;
  const int cbase = blockIdx.y * CC;
  const int tid = threadIdx.x, nthr = blockDim.x;

  for (int i = tid; i < 49 * LC; i += nthr) {
    const int tap = i / LC, l = i - tap * LC;
    *reinterpret_cast<float4*>(&wl[(flip ? 48 - tap : tap) * CC + l * 4]) =
        *reinterpret_cast<const float4*>(&w49c[tap * C + cbase + l * 4]);
  }
  // staging, batched: kSR rows x 2 units per thread are loaded before anything is stored, so 8 global loads are in
  // flight per thread (a one-load-per-iteration loop exposes the full memory latency 27x per workgroup)
  constexpr int kSR = 4;
  const int row_units = Wp * LC;
  for (int tr0 = 0; tr0 < TH + 6; tr0 += kSR) {
    for (int i0 = tid; i0 < row_units; i0 += 2 * nthr) {
      float4 v[kSR][2];
#pragma unroll
      for (int rr = 0; rr < kSR; ++rr) {
        const int hh = h0 - 3 + tr0 + rr;
        const bool row_in = hh >= 0 && hh < H && tr0 + rr < TH + 6;
        const TI* xr = x + ((n * H + (row_in ? hh : 0)) * static_cast<long>(W)) * C + cbase;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = i0 + k * nthr;
          const int col = i / LC, l = i - col * LC;
          const int ww = col - 3;
          v[rr][k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row_in && i < row_units && ww >= 0 && ww < W) v[rr][k] = load4(xr + static_cast<long>(ww) * C + l * 4);
        }
      }
#pragma unroll
      for (int rr = 0; rr < kSR; ++rr) {
        if (tr0 + rr >= TH + 6) continue;
        TS* trow = tile + static_cast<long>(tr0 + rr) * Wp * CC;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = i0 + k * nthr;
          if (i < row_units) lds_put4(trow + i * 4, v[rr][k]);          // unit i of a row = (col, l) at col*CC + l*4 = 4*i
        }
      }
    }
  }
  __syncthreads();

  const int n_sc = (W + kTW - 1) / kTW, n_sr = (TH + kRH - 1) / kRH;
  const int lc = tid % LC, sidx = tid / LC;
  if (sidx >= n_sr * n_sc) return;
  const int sc = sidx % n_sc, sr = sidx / n_sc;
  const int cl = lc * 4, c0 = cbase + cl;

  float4 acc[kRH][kTW];
  const float4 b4 = bias ? load4(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int oh = 0; oh < kRH; ++oh)
#pragma unroll
    for (int t = 0; t < kTW; ++t) acc[oh][t] = b4;

#pragma unroll
  for (int r = 0; r < kRH + 6; ++r) {
    const int tr = sr * kRH + r;
    if (tr >= TH + 6) continue;
    const TS* trow = tile + (static_cast<long>(tr) * Wp + sc * kTW) * CC + cl;
    float4 in[kTW + 6];
#pragma unroll
    for (int j = 0; j < kTW + 6; ++j)
      in[j] = (sc * kTW + j < Wp) ? load4(trow + j * CC) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int oh = 0; oh < kRH; ++oh) {
      const int kh = r - oh;
      if (kh < 0 || kh > 6) continue;
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const float4 wv = *reinterpret_cast<const float4*>(&wl[(kh * 7 + kw) * CC + cl]);
#pragma unroll
        for (int t = 0; t < kTW; ++t) fma4(acc[oh][t], wv, in[t + kw]);
      }
    }
  }
  const int h_end = min(H, h0 + TH);
#pragma unroll
  for (int oh = 0; oh < kRH; ++oh) {
    const int h = h0 + sr * kRH + oh;
    if (h >= h_end) continue;
#pragma unroll
    for (int t = 0; t < kTW; ++t) {
      const int w = sc * kTW + t;
      if (w >= W) continue;
      const long off = ((n * H + h) * static_cast<long>(W) + w) * C + c0;
      float4 v = acc[oh][t];
      if (add) { const float4 a = load4(add + off); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      store4(out + off, v);
    }
  }
}

// geometry of the tiled kernel for (H, W, C); returns false when the shape does not fit (caller uses the strip kernel)
struct DwTile { int cc, th, threads; size_t lds; };
inline bool dw_tile_plan(int H, int W, int C, size_t stage_elt, DwTile* t) {
  int cc = 0;
  if (C % 32 != 0) return false;
  if (C % 128 == 0 && W <= 8) cc = 128;
  else if (C % 64 == 0 && W <= 16) cc = 64;
  else cc = 32;
  const int n_sc = (W + kTW - 1) / kTW;
  int th = H;
  // rows per tile: as many as keep threads <= 256..512 and the tile under ~64 KiB (two workgroups per CU)
  auto lds_of = [&](int rows) { return static_cast<size_t>(49) * cc * 4 + static_cast<size_t>(rows + 6) * (W + 6) * cc * stage_elt; };
  auto thr_of = [&](int rows) { return (cc / 4) * ((rows + kRH - 1) / kRH) * n_sc; };
  while (th > 2 && (thr_of(th) > 512 || lds_of(th) > 72 * 1024)) th = (th + 1) / 2;
  th = (th + 1) & ~1;                                   // even number of rows (pairs of output rows per thread)
  if (th > H) th = H;
  if (thr_of(th) > 512 || lds_of(th) > 150 * 1024) return false;
  t->cc = cc; t->th = th; t->threads = ((thr_of(th) + 63) / 64) * 64; t->lds = lds_of(th);
  return true;
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 on bf16 operands with packed dot products (the autocast path: all ConvNeXt-T calls of the AT step).
//   49 fp32 FMAs per output make the stencil VALU-bound long before HBM (3.8 G FMA per call at 56x56x96, batch 256).
//   v_dot2c_f32_bf16 (acc += a.lo*b.lo + a.hi*b.hi, fp32 accumulate) halves that IF two W-adjacent inputs of ONE channel
//   share a dword — which NHWC memory does not give.  So the staging pass transposes pairs: the workgroup's zero-padded
//   tile is stored in LDS as  tile[row][pair m][channel] = (x[2m] | x[2m+1] << 16)  in padded column coordinates
//   (p = w + 3), one dword per channel.  A thread owns ONE channel (lane = channel: LDS reads are 32 consecutive dwords,
//   conflict-free) and a 4 x 8 output strip; per input row it reads 7 pair-dwords and issues 4 dot2 per output and
//   filter row:   even w = 2q  : pairs q..q+3 . {(f0,f1),(f2,f3),(f4,f5),(f6,0)}
//                 odd  w = 2q+1: pairs q..q+3 . {(0,f0),(f1,f2),(f3,f4),(f5,f6)}      (no realignment needed)
//   = 28 VALU per output instead of 49 + unpacking.  The 56 packed filter dwords of the lane's channel stay in registers.
//   Numerics = the reference's autocast convolution: bf16 inputs and weights, fp32 accumulation.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
constexpr int kDT = 8, kDR = 4, kDC = 32;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(uint16_t* p, float v) { *p = static_cast<uint16_t>(pack2_bf16(v, 0.f)); }

template <typename TI, typename TO>
__global__ __launch_bounds__(512) void dwconv7x7_dot2_kernel(const TI* __restrict__ x, const float* __restrict__ w49c,
                                                             const float* __restrict__ bias, const float* __restrict__ add,
                                                             TO* __restrict__ out, long N, int H, int W, int C, int flip,
                                                             int TH, int tiles_h, int ipw, int dbg) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile2[];       // [TH+6][P2][32]
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so with the plain
  // blockIdx -> tile map vertically adjacent tiles - which share 6 of their TH + 6 input rows - never meet in one L2 and
  // every halo row is fetched from HBM up to (TH + 6) / TH times.  Instead XCD x works through the x-th contiguous
  // eighth of the tile list, ordered (image, channel group, row tile) with the row tile fastest.
  long tsel;
  {
    const long L = blockIdx.x, B = gridDim.x;
    const long q = B / 8, r = B % 8, xcd = L % 8, k = L / 8;
    tsel = DW_DBG(dbg, 8) ? L : xcd * q + (xcd < r ? xcd : r) + k;          // dbg 8: timing experiment, plain order
  }
  const int n_cg = C / kDC;
  const int h0 = static_cast<int>(tsel % tiles_h) * TH;
  const int cbase = static_cast<int>((tsel / tiles_h) % n_cg) * kDC;
  const long ng = tsel / (static_cast<long>(tiles_h) * n_cg);     // group of ipw images
  const int tid = threadIdx.x, nthr = blockDim.x;

  // ---- this lane's channel: packed filter rows (rotated by 180 degrees when flip)
  const int lc = tid & (kDC - 1), sidx = tid / kDC;
  const int c = cbase + lc;
  uint32_t we[7][4], wo[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    float f[7];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int tap = kh * 7 + kw;
      f[kw] = w49c[(flip ? 48 - tap : tap) * C + c];
    }
    we[kh][0] = pack2_bf16(f[0], f[1]); we[kh][1] = pack2_bf16(f[2], f[3]); we[kh][2] = pack2_bf16(f[4], f[5]); we[kh][3] = pack2_bf16(f[6], 0.f);
    wo[kh][0] = pack2_bf16(0.f, f[0]); wo[kh][1] = pack2_bf16(f[1], f[2]); wo[kh][2] = pack2_bf16(f[3], f[4]); wo[kh][3] = pack2_bf16(f[5], f[6]);
  }
  // ---- ipw images per workgroup (whole-image tiles of the smallest maps; the filter registers above are set up once)
  for (int ii = 0; ii < ipw; ++ii) {
  const long n = ng * ipw + ii;
  if (n >= N) break;
  // ---- staging: unit = (row, pair m, 4-channel group): two 4-channel loads (padded cols 2m, 2m+1) -> 4 packed dwords
  constexpr int kSR = 4;
  const int row_units = P2 * (kDC / 4);
  // (all (row group, unit) pairs dealt over the whole workgroup: at 14x14 a row group is 88 units for 256 threads, and walking
  //  the five row groups one after the other exposed the load latency five times with two thirds of the threads idle)
  const int n_rg = (TH + 6 + kSR - 1) / kSR;
  {
    for (int j = tid; j < (DW_DBG(dbg, 2) ? 0 : n_rg * row_units); j += nthr) {   // dbg 2: timing experiment, no staging
      const int rg = j / row_units, i = j - rg * row_units, tr0 = rg * kSR;
      const int m = i / (kDC / 4), l4 = i - m * (kDC / 4);
      const int w0 = 2 * m - 3, w1 = w0 + 1;
      float4 v0[kSR], v1[kSR];
#pragma unroll
      for (int rr = 0; rr < kSR; ++rr) {
        const int hh = h0 - 3 + tr0 + rr;
        const bool row_in = hh >= 0 && hh < H && tr0 + rr < TH + 6;
        const TI* xr = x + ((n * H + (row_in ? hh : 0)) * static_cast<long>(W)) * C + cbase + l4 * 4;
        v0[rr] = (row_in && w0 >= 0 && w0 < W) ? load4(xr + static_cast<long>(w0) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
        v1[rr] = (row_in && w1 >= 0 && w1 < W) ? load4(xr + static_cast<long>(w1) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int rr = 0; rr < kSR; ++rr) {
        if (tr0 + rr >= TH + 6) continue;
        uint4 d;
        d.x = pack2_bf16(v0[rr].x, v1[rr].x); d.y = pack2_bf16(v0[rr].y, v1[rr].y);
        d.z = pack2_bf16(v0[rr].z, v1[rr].z); d.w = pack2_bf16(v0[rr].w, v1[rr].w);
        *reinterpret_cast<uint4*>(&tile2[(static_cast<long>(tr0 + rr) * P2 + m) * kDC + l4 * 4]) = d;
      }
    }
  }

  __syncthreads();

  const int n_sr = (TH + kDR - 1) / kDR;
  const bool worker = sidx < n_sr * n_sc;                       // the block is rounded up to whole wavefronts
  const int sc = worker ? sidx % n_sc : 0, sr = worker ? sidx / n_sc : 0;
  float acc[kDR][kDT];
  const float b0 = bias ? bias[c] : 0.f;
  const int h_end = min(H, h0 + TH);
#pragma unroll
  for (int oh = 0; oh < kDR; ++oh)
#pragma unroll
    for (int t = 0; t < kDT; ++t) acc[oh][t] = b0;

  if (worker) {
#pragma unroll
    for (int r = 0; r < kDR + 6; ++r) {
      const int tr = sr * kDR + r;
      if (tr >= TH + 6 || DW_DBG(dbg, 1)) continue;                           // dbg 1: timing experiment, no stencil arithmetic
      const uint32_t* trow = tile2 + (static_cast<long>(tr) * P2 + sc * (kDT / 2)) * kDC + lc;
      uint32_t d[kDT / 2 + 3];
#pragma unroll
      for (int i = 0; i < kDT / 2 + 3; ++i) d[i] = trow[i * kDC];
#pragma unroll
      for (int oh = 0; oh < kDR; ++oh) {
        const int kh = r - oh;
        if (kh < 0 || kh > 6) continue;
#pragma unroll
        for (int q = 0; q < kDT / 2; ++q) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[oh][2 * q] = dot2(d[q + i], we[kh][i], acc[oh][2 * q]);
            acc[oh][2 * q + 1] = dot2(d[q + i], wo[kh][i], acc[oh][2 * q + 1]);
          }
        }
      }
    }
  }
  // ---- epilogue through LDS: lane = channel during the stencil means 2..4-byte accesses per lane (32 channels = one
  //      64..128-byte run per instruction and position); transposed through the (now dead) input tile every thread moves 16
  //      bytes of the NHWC tensor per instruction instead - for the result AND for the fused "+ add" operand
  __syncthreads();
  TO* ot = reinterpret_cast<TO*>(tile2);                                 // [rows][W][32]
  if (worker) {
#pragma unroll
    for (int oh = 0; oh < kDR; ++oh) {
      const int hl = sr * kDR + oh;
#pragma unroll
      for (int t = 0; t < kDT; ++t) {
        const int w = sc * kDT + t;
        if (h0 + hl < h_end && w < W) store1(ot + (static_cast<long>(hl) * W + w) * kDC + lc, acc[oh][t]);
      }
    }
  }
  __syncthreads();
  constexpr int EPC = 16 / static_cast<int>(sizeof(TO));                 // elements per 16-byte chunk
  constexpr int CH = kDC / EPC;                                          // chunks per position
  const int n_chunks = (h_end - h0) * W * CH;
  for (int i = tid; i < n_chunks; i += nthr) {
    const int pos = i / CH, ch = i - pos * CH;
    const long off = ((n * H + h0) * static_cast<long>(W) + pos) * C + cbase + ch * EPC;
    uint4 v = *reinterpret_cast<const uint4*>(ot + static_cast<long>(pos) * kDC + ch * EPC);
    if constexpr (sizeof(TO) == 4) {
      if (add) {
        const float4 a4 = *reinterpret_cast<const float4*>(add + off);
        float4 f = __builtin_bit_cast(float4, v);
        f.x += a4.x; f.y += a4.y; f.z += a4.z; f.w += a4.w;
        v = __builtin_bit_cast(uint4, f);
      }
    }
    *reinterpret_cast<uint4*>(out + off) = v;
  }
  __syncthreads();                                                       // the output tile aliases the next image's input tile
  }
}

// ------------------------------------------------------------------------------------------------
// Rolling-window variant for the large maps (56x56, 28x28).  The tile kernel above stages TH + 6 input rows for TH = 4
// output rows (2.5x the tensor through the load / pack / LDS-write path, ~1/3 of its time) and sets its 56 filter dwords
// up once per 4 rows.  Here a workgroup walks DOWN a band of an image: the 10-row window lives in LDS as a ring, each
// iteration adds only the 4 new rows - loaded into registers BEFORE the arithmetic of the current rows (and so is the
// fused "+ add" operand), written to the ring after it - and the filter registers are set up once per band.
//   LDS: ring [10][P2][32] dwords + output tile [4][W][32] TO (transposed epilogue: 16 bytes per lane).
// ------------------------------------------------------------------------------------------------
constexpr int kRollMaxThreads = 384;                  // 10 column strips of 8 x 32 channels (maps up to 80 pixels wide, ConvNeXt-L @320) + staging-only lanes
// MAXT: the launch bound is part of the instantiation - 256 threads (maps up to 64 pixels wide: every ConvNeXt-T / -B / ViT
// shape) - and only the 9 / 10-strip launches of the 65 ... 80-pixel maps are built with the wider bound, always with ONE staging
// unit per thread (a sixth, staging-only wavefront instead of two units on five): round 2 had one 320-thread bound for all
// and its U = 2 kernels - the 28x28 maps of the headline configuration among them - carried 21-22 spilled VGPRs.
template <typename TI, typename TO, int U, bool FA = false, int MAXT = 256>
__global__ __launch_bounds__(MAXT) void dwconv7x7_roll_kernel(const TI* __restrict__ x, const float* __restrict__ w49c,
                                                             const float* __restrict__ bias, const float* __restrict__ add,
                                                             TO* __restrict__ out, int H, int W, int C, int flip, int RS,
                                                             int n_seg) {
  constexpr int TH = kDR, WR = kDR + 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t tile2[];
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  uint32_t* win = tile2;                                                  // [WR][P2][32]
  using TS = typename std::conditional<FA, float, TO>::type;              // FA: fp32 staging so that "+ add" is summed before the one rounding to bf16
  TS* ot = reinterpret_cast<TS*>(tile2 + WR * P2 * kDC);                  // [TH][W][32]
  const int n_cg = C / kDC;
  long b = blockIdx.x;
  const int seg = static_cast<int>(b % n_seg);
  b /= n_seg;
  const int cbase = static_cast<int>(b % n_cg) * kDC;
  const long n = b / n_cg;
  const int r_begin = seg * RS, r_end = min(H, r_begin + RS);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row_units = P2 * (kDC / 4);

  // staging unit = (pair m, 4-channel group): padded columns 2m, 2m+1 -> 4 packed dwords
  using Raw = typename std::conditional<sizeof(TI) == 4, float4, uint2>::type;
  int um[U], ul4[U];
  bool uok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = min(tid + u * nthr, row_units - 1);
    uok[u] = tid + u * nthr < row_units;
    um[u] = i / (kDC / 4);
    ul4[u] = i - um[u] * (kDC / 4);
  }
  // Loads are unconditional (clamped coordinates, zero selected afterwards): a guarded load compiles to a branch with
  // its own s_waitcnt, which serialises the round trips this kernel exists to overlap.
  auto load_row = [&](int hh, Raw* v0, Raw* v1) {
    const int hc = min(max(hh, 0), H - 1);
    const TI* xrow = x + ((n * H + hc) * static_cast<long>(W)) * C + cbase;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w0 = 2 * um[u] - 3, w1 = w0 + 1;
      const TI* xr = xrow + ul4[u] * 4;
      const Raw r0 = *reinterpret_cast<const Raw*>(xr + static_cast<long>(min(max(w0, 0), W - 1)) * C);
      const Raw r1 = *reinterpret_cast<const Raw*>(xr + static_cast<long>(min(max(w1, 0), W - 1)) * C);
      v0[u] = r0;                                       // zeroing of the padding happens in store_row: a select here would
      v1[u] = r1;                                       // wait for the load on the spot
    }
  };
  auto store_row = [&](int slot, int hh, const Raw* v0, const Raw* v1) {
    const bool row_in = hh >= 0 && hh < H;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!uok[u]) continue;
      const int w0 = 2 * um[u] - 3, w1 = w0 + 1;
      const bool ok0 = row_in && w0 >= 0 && w0 < W, ok1 = row_in && w1 >= 0 && w1 < W;
      uint4 d;
      if constexpr (sizeof(TI) == 4) {
        d.x = pack2_bf16(v0[u].x, v1[u].x); d.y = pack2_bf16(v0[u].y, v1[u].y);
        d.z = pack2_bf16(v0[u].z, v1[u].z); d.w = pack2_bf16(v0[u].w, v1[u].w);
      } else {                                                            // bf16 input: interleave the two columns' halves
        d.x = (v0[u].x & 0xffffu) | (v1[u].x << 16); d.y = (v0[u].x >> 16) | (v1[u].x & 0xffff0000u);
        d.z = (v0[u].y & 0xffffu) | (v1[u].y << 16); d.w = (v0[u].y >> 16) | (v1[u].y & 0xffff0000u);
      }
      const uint32_t k0 = ok0 ? 0x0000ffffu : 0u, k1 = ok1 ? 0xffff0000u : 0u, km = k0 | k1;
      d.x &= km; d.y &= km; d.z &= km; d.w &= km;
      *reinterpret_cast<uint4*>(&win[(static_cast<long>(slot) * P2 + um[u]) * kDC + ul4[u] * 4]) = d;
    }
  };

  // ---- fill the ring: input rows r_begin-3 .. r_begin+6 -> slots 0..9
  {
    Raw v0[TH][U], v1[TH][U];
#pragma unroll
    for (int r0 = 0; r0 < WR; r0 += TH) {
#pragma unroll
      for (int rr = 0; rr < TH; ++rr)
        if (r0 + rr < WR) load_row(r_begin - 3 + r0 + rr, v0[rr], v1[rr]);
#pragma unroll
      for (int rr = 0; rr < TH; ++rr)
        if (r0 + rr < WR) store_row(r0 + rr, r_begin - 3 + r0 + rr, v0[rr], v1[rr]);
    }
  }

  // ---- this lane's channel: packed filter rows (rotated by 180 degrees when flip)
  const int lc = tid & (kDC - 1), sidx = tid / kDC;
  const int c = cbase + lc;
  uint32_t we[7][4], wo[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    float f[7];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int tap = kh * 7 + kw;
      f[kw] = w49c[(flip ? 48 - tap : tap) * C + c];
    }
    we[kh][0] = pack2_bf16(f[0], f[1]); we[kh][1] = pack2_bf16(f[2], f[3]); we[kh][2] = pack2_bf16(f[4], f[5]); we[kh][3] = pack2_bf16(f[6], 0.f);
    wo[kh][0] = pack2_bf16(0.f, f[0]); wo[kh][1] = pack2_bf16(f[1], f[2]); wo[kh][2] = pack2_bf16(f[3], f[4]); wo[kh][3] = pack2_bf16(f[5], f[6]);
  }
  const float b0 = bias ? bias[c] : 0.f;
  const bool worker = sidx < n_sc;
  const int sc = worker ? sidx : 0;
  constexpr int EPC = 16 / static_cast<int>(sizeof(TS));                 // elements per 16-byte chunk
  constexpr int CH = kDC / EPC;                                          // chunks per position
  constexpr int AC = 8;                                                  // "+ add" chunks prefetched per thread (fp32 output)
  __syncthreads();

  int s0 = 0;                                                            // ring slot of input row h0 - 3
  for (int h0 = r_begin; h0 < r_end; h0 += TH) {
    const int h_end = min(r_end, h0 + TH);
    const bool more = h0 + TH < r_end;
    // ---- global loads of the NEXT iteration's rows and of this iteration's add operand: in flight during the arithmetic
    Raw p0[TH][U], p1[TH][U];
    if (more) {
#pragma unroll
      for (int rr = 0; rr < TH; ++rr) load_row(h0 + TH + 3 + rr, p0[rr], p1[rr]);
    }
    const int n_chunks = (h_end - h0) * W * CH;
    const long obase = ((n * H + h0) * static_cast<long>(W)) * C + cbase;
    static_assert(MAXT <= 256 || U == 1, "wide launches stage one unit per thread");
    float4 a4[AC];
    if constexpr (sizeof(TS) == 4) {
      if (add) {
#pragma unroll
        for (int k = 0; k < AC; ++k) {
          const int i = min(tid + k * nthr, n_chunks - 1);                 // clamped: the load stays unconditional
          const int pos = i / CH, ch = i - pos * CH;
          a4[k] = *reinterpret_cast<const float4*>(add + obase + static_cast<long>(pos) * C + ch * EPC);
        }
      }
    }
    // ---- stencil on the ring
    if (worker) {
      float acc[kDR][kDT];
#pragma unroll
      for (int oh = 0; oh < kDR; ++oh)
#pragma unroll
        for (int t = 0; t < kDT; ++t) acc[oh][t] = b0;
#pragma unroll
      for (int r = 0; r < kDR + 6; ++r) {
        int slot = s0 + r;
        if (slot >= WR) slot -= WR;
        const uint32_t* trow = win + (static_cast<long>(slot) * P2 + sc * (kDT / 2)) * kDC + lc;
        uint32_t d[kDT / 2 + 3];
#pragma unroll
        for (int i = 0; i < kDT / 2 + 3; ++i) d[i] = trow[i * kDC];
#pragma unroll
        for (int oh = 0; oh < kDR; ++oh) {
          const int kh = r - oh;
          if (kh < 0 || kh > 6) continue;
#pragma unroll
          for (int q = 0; q < kDT / 2; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[oh][2 * q] = dot2(d[q + i], we[kh][i], acc[oh][2 * q]);
              acc[oh][2 * q + 1] = dot2(d[q + i], wo[kh][i], acc[oh][2 * q + 1]);
            }
          }
        }
      }
#pragma unroll
      for (int oh = 0; oh < kDR; ++oh)
#pragma unroll
        for (int t = 0; t < kDT; ++t) {
          const int w = sc * kDT + t;
          if (h0 + oh < h_end && w < W) store1(ot + (static_cast<long>(oh) * W + w) * kDC + lc, acc[oh][t]);
        }
    }
    __syncthreads();                                                     // output tile complete; ring rows free to replace
    // ---- transposed epilogue: 16 bytes per lane
#pragma unroll
    for (int k = 0; k < AC; ++k) {
      const int i = tid + k * nthr;
      if (i >= n_chunks) break;
      const int pos = i / CH, ch = i - pos * CH;
      uint4 v = *reinterpret_cast<const uint4*>(ot + static_cast<long>(pos) * kDC + ch * EPC);
      if constexpr (sizeof(TS) == 4) {
        if (add) {
          float4 f = __builtin_bit_cast(float4, v);
          f.x += a4[k].x; f.y += a4[k].y; f.z += a4[k].z; f.w += a4[k].w;
          v = __builtin_bit_cast(uint4, f);
        }
      }
      if constexpr (sizeof(TS) == sizeof(TO)) {
        *reinterpret_cast<uint4*>(out + obase + static_cast<long>(pos) * C + ch * EPC) = v;
      } else {                                                           // fp32 sum -> bf16 result, rounded once
        const float4 f = __builtin_bit_cast(float4, v);
        *reinterpret_cast<uint2*>(out + obase + static_cast<long>(pos) * C + ch * EPC) = make_uint2(pack2_bf16(f.x, f.y), pack2_bf16(f.z, f.w));
      }
    }
    // ---- the prefetched rows replace the 4 oldest ring rows
    if (more) {
#pragma unroll
      for (int rr = 0; rr < TH; ++rr) {
        int slot = s0 + rr;
        if (slot >= WR) slot -= WR;
        store_row(slot, h0 + TH + 3 + rr, p0[rr], p1[rr]);
      }
    }
    s0 += TH;
    if (s0 >= WR) s0 -= WR;
    __syncthreads();
  }
}

struct DwRoll { int threads, units, rs, n_seg; size_t lds; };
inline bool dw_roll_plan(int H, int W, int C, int in_bytes, int out_bytes, DwRoll* t) {
  constexpr int off = 1;          // 0: tile kernel everywhere
  // smallest map (pixels).  28x28 and up always; 20x20 ... 27x27 (ConvNeXt-L @320 stage 2, 768 channels, batch 128; too many
  // strips for the multi-image kernel) measured 123 -> 107 us forward and 235 -> 125 us input gradient + add against the
  // whole-image tile kernel; at 10x10 the tile / multi-image kernels stay ahead for fp32 inputs.
  constexpr int roll_min = 400;
  if (!off || C % kDC != 0 || H * W < roll_min) return false;
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  constexpr int max_sc = kRollMaxThreads / kDC;
  if (n_sc > max_sc || n_sc * kDC > kRollMaxThreads) return false;
  // 65 ... 80 pixels wide (ConvNeXt-L @320 stage 0, 80x80x192, batch 128): five wavefronts and up to 96 KB of LDS leave one
  // workgroup per CU - still ahead of the tile kernels for bf16 inputs (697 -> 441 us all-bf16, 897 -> 792 us input gradient
  // + add), behind them for fp32 inputs (503 -> 559 us)
  if (n_sc > 8 && in_bytes != 2) return false;
  t->threads = ((n_sc * kDC + 63) / 64) * 64;      // (extra staging-only wavefronts were tried at 28x28: fewer workgroups fit, slower)
  if (n_sc > 8) t->threads = std::min(kRollMaxThreads, ((P2 * (kDC / 4) + 63) / 64) * 64);   // wide maps: one staging unit per thread
  t->units = (P2 * (kDC / 4) + t->threads - 1) / t->threads;
  if (t->units > 2 || (t->threads > 256 && t->units != 1)) return false;
  const int chunks = kDR * W * (kDC * out_bytes / 16);
  if ((chunks + t->threads - 1) / t->threads > 8) return false;                                // AC in the kernel
  constexpr int rs_env = 0;            // tuning experiments only
  // band height, measured (tools/dw_bench.py): 28 rows at 56x56 (two bands per image) and for fp32 tensors at 28x28, 16 for all-bf16 28x28
  int rs = rs_env > 0 ? rs_env : ((H >= 56 || in_bytes == 4 || out_bytes == 4) ? 28 : 16);
  rs = ((rs + kDR - 1) / kDR) * kDR;
  t->rs = rs;
  t->n_seg = (H + rs - 1) / rs;
  t->lds = static_cast<size_t>(kDR + 6) * P2 * kDC * 4 + static_cast<size_t>(kDR) * W * kDC * out_bytes;
  return t->lds <= 150 * 1024;
}

// ------------------------------------------------------------------------------------------------
// Small maps (14x14, 7x7): the whole zero-padded image x 32 channels is one tile.  A workgroup walks over IPW images of one
// channel group with the same software pipeline as the rolling kernel: image i+1 (and the add operand of image i) is
// loaded into registers before the stencil of image i runs, the filter registers are set up once.
//   LDS: tile [TH+6][P2][32] dwords (border rows / columns zeroed once) + output tile [H][W][32] TO.
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO, int UT, bool FA = false>
__global__ __launch_bounds__(256) void dwconv7x7_multi_kernel(const TI* __restrict__ x, const float* __restrict__ w49c,
                                                              const float* __restrict__ bias, const float* __restrict__ add,
                                                              TO* __restrict__ out, int N, int H, int W, int C, int flip,
                                                              int n_sr, int ipw) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile2[];
  using Raw = typename std::conditional<sizeof(TI) == 4, float4, uint2>::type;
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  const int TR = n_sr * kDR + 6;                                          // tile rows
  uint32_t* win = tile2;
  using TS = typename std::conditional<FA, float, TO>::type;              // FA: fp32 staging so that "+ add" is summed before the one rounding to bf16
  TS* ot = reinterpret_cast<TS*>(tile2 + TR * P2 * kDC);                  // [H][W][32]
  const int n_cg = C / kDC;
  const int cbase = static_cast<int>(blockIdx.x % n_cg) * kDC;
  const long n_first = static_cast<long>(blockIdx.x / n_cg) * ipw;
  const long n_last = min(static_cast<long>(N), n_first + ipw);
  const int tid = threadIdx.x, nthr = blockDim.x;
  // staging unit = (row h, pair m covering columns 2m-3, 2m-2, 4-channel group); only pairs that touch real columns
  const int m_lo = 1, m_hi = (W + 2) / 2 + 1;                             // pairs m_lo .. m_hi overlap [0, W)
  const int upr = (m_hi - m_lo + 1) * (kDC / 4), total_units = H * upr;
  int uh[UT], um[UT], ul4[UT];
  bool uok[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int i = min(tid + u * nthr, total_units - 1);
    uok[u] = tid + u * nthr < total_units;
    uh[u] = i / upr;
    const int j = i - uh[u] * upr;
    um[u] = m_lo + j / (kDC / 4);
    ul4[u] = j % (kDC / 4);
  }
  auto load_img = [&](long n, Raw* v0, Raw* v1) {
    const TI* xi = x + (n * H) * static_cast<long>(W) * C + cbase;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int w0 = 2 * um[u] - 3, w1 = w0 + 1;
      const TI* xr = xi + (static_cast<long>(uh[u]) * W) * C + ul4[u] * 4;
      v0[u] = *reinterpret_cast<const Raw*>(xr + static_cast<long>(min(max(w0, 0), W - 1)) * C);
      v1[u] = *reinterpret_cast<const Raw*>(xr + static_cast<long>(min(max(w1, 0), W - 1)) * C);
    }
  };
  auto store_img = [&](const Raw* v0, const Raw* v1) {
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (!uok[u]) continue;
      const int w0 = 2 * um[u] - 3, w1 = w0 + 1;
      uint4 d;
      if constexpr (sizeof(TI) == 4) {
        d.x = pack2_bf16(v0[u].x, v1[u].x); d.y = pack2_bf16(v0[u].y, v1[u].y);
        d.z = pack2_bf16(v0[u].z, v1[u].z); d.w = pack2_bf16(v0[u].w, v1[u].w);
      } else {
        d.x = (v0[u].x & 0xffffu) | (v1[u].x << 16); d.y = (v0[u].x >> 16) | (v1[u].x & 0xffff0000u);
        d.z = (v0[u].y & 0xffffu) | (v1[u].y << 16); d.w = (v0[u].y >> 16) | (v1[u].y & 0xffff0000u);
      }
      const uint32_t km = ((w0 >= 0 && w0 < W) ? 0x0000ffffu : 0u) | ((w1 >= 0 && w1 < W) ? 0xffff0000u : 0u);
      d.x &= km; d.y &= km; d.z &= km; d.w &= km;
      *reinterpret_cast<uint4*>(&win[(static_cast<long>(uh[u] + 3) * P2 + um[u]) * kDC + ul4[u] * 4]) = d;
    }
  };

  // ---- zero the tile once (the staging only ever rewrites the pairs that overlap the image), stage the first image
  for (int i = tid; i < TR * P2 * kDC / 4; i += nthr) reinterpret_cast<uint4*>(win)[i] = make_uint4(0u, 0u, 0u, 0u);
  Raw p0[UT], p1[UT];
  load_img(n_first, p0, p1);
  const int lc = tid & (kDC - 1), sidx = tid / kDC;
  const int c = cbase + lc;
  uint32_t we[7][4], wo[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    float f[7];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int tap = kh * 7 + kw;
      f[kw] = w49c[(flip ? 48 - tap : tap) * C + c];
    }
    we[kh][0] = pack2_bf16(f[0], f[1]); we[kh][1] = pack2_bf16(f[2], f[3]); we[kh][2] = pack2_bf16(f[4], f[5]); we[kh][3] = pack2_bf16(f[6], 0.f);
    wo[kh][0] = pack2_bf16(0.f, f[0]); wo[kh][1] = pack2_bf16(f[1], f[2]); wo[kh][2] = pack2_bf16(f[3], f[4]); wo[kh][3] = pack2_bf16(f[5], f[6]);
  }
  const float b0 = bias ? bias[c] : 0.f;
  const bool worker = sidx < n_sr * n_sc;
  const int sc = worker ? sidx % n_sc : 0, sr = worker ? sidx / n_sc : 0;
  constexpr int EPC = 16 / static_cast<int>(sizeof(TS));
  constexpr int CH = kDC / EPC;
  constexpr int AC = 8;
  const int n_chunks = H * W * CH;
  __syncthreads();                                                        // zero fill done
  store_img(p0, p1);
  __syncthreads();

  for (long n = n_first; n < n_last; ++n) {
    const bool more = n + 1 < n_last;
    if (more) load_img(n + 1, p0, p1);
    const long obase = (n * H) * static_cast<long>(W) * C + cbase;
    float4 a4[AC];
    if constexpr (sizeof(TS) == 4) {
      if (add) {
#pragma unroll
        for (int k = 0; k < AC; ++k) {
          const int i = min(tid + k * nthr, n_chunks - 1);
          const int pos = i / CH, ch = i - pos * CH;
          a4[k] = *reinterpret_cast<const float4*>(add + obase + static_cast<long>(pos) * C + ch * EPC);
        }
      }
    }
    if (worker) {
      float acc[kDR][kDT];
#pragma unroll
      for (int oh = 0; oh < kDR; ++oh)
#pragma unroll
        for (int t = 0; t < kDT; ++t) acc[oh][t] = b0;
#pragma unroll
      for (int r = 0; r < kDR + 6; ++r) {
        const uint32_t* trow = win + (static_cast<long>(sr * kDR + r) * P2 + sc * (kDT / 2)) * kDC + lc;
        uint32_t d[kDT / 2 + 3];
#pragma unroll
        for (int i = 0; i < kDT / 2 + 3; ++i) d[i] = trow[i * kDC];
#pragma unroll
        for (int oh = 0; oh < kDR; ++oh) {
          const int kh = r - oh;
          if (kh < 0 || kh > 6) continue;
#pragma unroll
          for (int q = 0; q < kDT / 2; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[oh][2 * q] = dot2(d[q + i], we[kh][i], acc[oh][2 * q]);
              acc[oh][2 * q + 1] = dot2(d[q + i], wo[kh][i], acc[oh][2 * q + 1]);
            }
          }
        }
      }
#pragma unroll
      for (int oh = 0; oh < kDR; ++oh)
#pragma unroll
        for (int t = 0; t < kDT; ++t) {
          const int h = sr * kDR + oh, w = sc * kDT + t;
          if (h < H && w < W) store1(ot + (static_cast<long>(h) * W + w) * kDC + lc, acc[oh][t]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < AC; ++k) {
      const int i = tid + k * nthr;
      if (i >= n_chunks) break;
      const int pos = i / CH, ch = i - pos * CH;
      uint4 v = *reinterpret_cast<const uint4*>(ot + static_cast<long>(pos) * kDC + ch * EPC);
      if constexpr (sizeof(TS) == 4) {
        if (add) {
          float4 f = __builtin_bit_cast(float4, v);
          f.x += a4[k].x; f.y += a4[k].y; f.z += a4[k].z; f.w += a4[k].w;
          v = __builtin_bit_cast(uint4, f);
        }
      }
      if constexpr (sizeof(TS) == sizeof(TO)) {
        *reinterpret_cast<uint4*>(out + obase + static_cast<long>(pos) * C + ch * EPC) = v;
      } else {                                                           // fp32 sum -> bf16 result, rounded once
        const float4 f = __builtin_bit_cast(float4, v);
        *reinterpret_cast<uint2*>(out + obase + static_cast<long>(pos) * C + ch * EPC) = make_uint2(pack2_bf16(f.x, f.y), pack2_bf16(f.z, f.w));
      }
    }
    if (more) store_img(p0, p1);
    __syncthreads();
  }
}

struct DwMulti { int threads, n_sr, ut, ipw; size_t lds; };
inline bool dw_multi_plan(int N, int H, int W, int C, int in_bytes, int out_bytes, DwMulti* t) {
  constexpr int on = 1;          // 0: tile kernel
  // measured (tools/dw_bench.py, 14x14x384 / 7x7x768): bf16 inputs 74 -> 50 us / 42 -> 31 us, fp32 inputs no better than the
  // tile kernel (40 prefetch registers per thread at the 256-VGPR limit) - those stay there
  if (!on || C % kDC != 0 || H * W >= 784 || in_bytes == 4) return false;
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  const int n_sr = (H + kDR - 1) / kDR;
  if (n_sr * n_sc * kDC > 256) return false;
  t->n_sr = n_sr;
  t->threads = ((n_sr * n_sc * kDC + 63) / 64) * 64;
  const int upr = ((W + 2) / 2 + 1) * (kDC / 4);
  const int ut = (H * upr + t->threads - 1) / t->threads;
  t->ut = ut <= 5 ? 5 : (ut <= 7 ? 7 : 0);
  if (!t->ut) return false;
  if ((H * W * (kDC * out_bytes / 16) + t->threads - 1) / t->threads > 8) return false;        // AC in the kernel
  constexpr int ipw_env = 0;         // tuning experiments only
  t->ipw = ipw_env > 0 ? ipw_env : (H * W <= 64 ? 2 : 4);
  t->lds = static_cast<size_t>(n_sr * kDR + 6) * P2 * kDC * 4 + static_cast<size_t>(H) * W * kDC * out_bytes;
  return t->lds <= 150 * 1024;
}

struct DwDot { int th, threads; size_t lds; };
inline bool dw_dot2_plan(int H, int W, int C, DwDot* t) {
  if (C % kDC != 0) return false;
  const int n_sc = (W + kDT - 1) / kDT, P2 = n_sc * (kDT / 2) + 3;
  if (n_sc > 16) return false;
  int max_sr = 16 / n_sc;                                  // <= 512 threads
  int th = H < kDR * max_sr ? H : kDR * max_sr;
  // large maps (56x56, 28x28): 4-row tiles - a 40 KiB tile lets 3-4 workgroups share a CU, so one workgroup's staging and
  // store phases overlap another's arithmetic (measured 508 -> 414 us for the 56x56x96 input gradient); small maps are
  // taken whole
  if (H * W >= 784 && th > kDR) th = kDR;
  while (th > kDR && static_cast<size_t>(th + 6) * P2 * kDC * 4 > 64 * 1024) th -= kDR;
  const int n_sr = (th + kDR - 1) / kDR;
  t->th = th; t->threads = ((n_sr * n_sc * kDC + 63) / 64) * 64;
  constexpr int th_override = 0;     // tuning experiments only
  if (th_override > 0 && th_override <= th) {
    t->th = th_override;
    t->threads = ((((th_override + kDR - 1) / kDR) * n_sc * kDC + 63) / 64) * 64;
    t->lds = static_cast<size_t>(th_override + 6) * P2 * kDC * 4;
    if (t->lds < static_cast<size_t>(th_override) * W * kDC * 4) t->lds = static_cast<size_t>(th_override) * W * kDC * 4;
    return true;
  }
  t->lds = static_cast<size_t>(th + 6) * P2 * kDC * 4;
  // the transposed output tile [th][W][32] aliases the input tile: with fp32 results it is the larger of the two once
  // th * W > (th + 6) * P2 (e.g. a whole 20x20 map: 51 200 vs 49 920 bytes - found by the ConvNeXt-L @320 oracle test)
  if (t->lds < static_cast<size_t>(th) * W * kDC * 4) t->lds = static_cast<size_t>(th) * W * kDC * 4;
  return t->lds <= 150 * 1024;
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 filter gradient.  block = (LC, 7): threadIdx.y = kh, each thread owns 7 taps x 4
// channels of fp32 accumulators and walks strips (1 row x TW columns) with a grid-stride loop;
// per-block partials go to ws[block][50][C] (tap 49 = bias gradient) and are summed in fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int kWgradBlocks = 512;

template <typename TX, typename TD>
__global__ __launch_bounds__(7 * 48) void dwconv7x7_wgrad_kernel(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                                 float* __restrict__ ws, int H, int W, int C,
                                                                 int tiles_w, long n_strips) {
  const int cbase = blockIdx.y * kCC;
  const int cc = min(kCC, C - cbase);
  const int cl = threadIdx.x * 4, kh = threadIdx.y;
  if (cl >= cc) return;
  const int c0 = cbase + cl;
  float4 acc[7], accb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long s = blockIdx.x; s < n_strips; s += gridDim.x) {
    const int tw = static_cast<int>(s % tiles_w);
    const long r1 = s / tiles_w;
    const int h = static_cast<int>(r1 % H);
    const long n = r1 / H;
    const int w0 = tw * kTW;
    const int hh = h + kh - 3;
    float4 d[kTW];
    const TD* dr = dy + ((n * H + h) * static_cast<long>(W)) * C + c0;
#pragma unroll
    for (int t = 0; t < kTW; ++t)
      d[t] = (w0 + t < W) ? load4(dr + static_cast<long>(w0 + t) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (kh == 0) {
#pragma unroll
      for (int t = 0; t < kTW; ++t) { accb.x += d[t].x; accb.y += d[t].y; accb.z += d[t].z; accb.w += d[t].w; }
    }
    if (hh < 0 || hh >= H) continue;
    const TX* xr = x + ((n * H + hh) * static_cast<long>(W)) * C + c0;
    float4 in[kTW + 6];
#pragma unroll
    for (int j = 0; j < kTW + 6; ++j) {
      const int col = w0 - 3 + j;
      in[j] = (col >= 0 && col < W) ? load4(xr + static_cast<long>(col) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kw = 0; kw < 7; ++kw)
#pragma unroll
      for (int t = 0; t < kTW; ++t) fma4(acc[kw], d[t], in[t + kw]);
  }
  float* p = ws + static_cast<long>(blockIdx.x) * 50 * C;
#pragma unroll
  for (int kw = 0; kw < 7; ++kw) store4(p + (kh * 7 + kw) * C + c0, acc[kw]);
  if (kh == 0) store4(p + 49 * C + c0, accb);
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x7 filter gradient on bf16 operands with packed dot products (autocast path).
//   dw[c][kh][kw] = sum_{n,h,w} dy[n,h,w,c] * x[n,h+kh-3,w+kw-3,c]          db[c] = sum dy
//   Same LDS image as the forward (pairs of W-adjacent values of one channel per dword, padded coordinates p = w+3),
//   plus the dy tile in pair form.  A thread owns (channel = lane, filter row kh) and walks the tile's output rows:
//   per pair step m it reads one dy pair and one new x pair, rebuilds the three odd-offset pairs with v_alignbit and
//   issues 7 dot2 (one per kw):   even kw = 2j : pair m+j          odd kw = 2j+1 : (hi(pair m+j), lo(pair m+j+1)).
//   A workgroup = (image group, 32-channel chunk); it keeps its 49 x 32 partial sums in registers over all of its
//   tiles and writes them once to ws[part][50][C] (tap 49 = bias gradient); reduce_parts_kernel sums the parts in a
//   fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
template <typename TX, typename TD>
__global__ __launch_bounds__(256) void dwconv7x7_wgrad_dot2_kernel(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                                   float* __restrict__ ws, long N, int H, int W, int C,
                                                                   int TH, int tiles_h, int imgs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t wg_lds[];
  const int D2 = (W + 1) / 2, P2 = D2 + 3;
  uint32_t* xt = wg_lds;                                         // [TH+6][P2][32]
  uint32_t* dt = wg_lds + static_cast<long>(TH + 6) * P2 * kDC;  // [TH][D2][32]
  const int cbase = blockIdx.y * kDC;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lc = tid & (kDC - 1), kh = tid / kDC;                // kh == 7: staging helper only

  float acc[7], accb = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = 0.f;
  const uint32_t ones = 0x3f803f80u;                             // (1.0bf16, 1.0bf16)

  const long n_begin = static_cast<long>(blockIdx.x) * imgs;
  const long n_end = n_begin + imgs < N ? n_begin + imgs : N;
  for (long n = n_begin; n < n_end; ++n) {
    for (int th = 0; th < tiles_h; ++th) {
      const int h0 = th * TH;
      // ---- stage x (zero-padded, pairs) and dy (pairs)
      const int xu = P2 * (kDC / 4);
      for (int tr0 = 0; tr0 < TH + 6; tr0 += 4) {
        for (int i = tid; i < xu; i += nthr) {
          const int m = i / (kDC / 4), l4 = i - m * (kDC / 4);
          const int w0 = 2 * m - 3, w1 = w0 + 1;
          float4 v0[4], v1[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int hh = h0 - 3 + tr0 + rr;
            const bool row_in = hh >= 0 && hh < H && tr0 + rr < TH + 6;
            const TX* xr = x + ((n * H + (row_in ? hh : 0)) * static_cast<long>(W)) * C + cbase + l4 * 4;
            v0[rr] = (row_in && w0 >= 0 && w0 < W) ? load4(xr + static_cast<long>(w0) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
            v1[rr] = (row_in && w1 >= 0 && w1 < W) ? load4(xr + static_cast<long>(w1) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (tr0 + rr >= TH + 6) continue;
            uint4 d;
            d.x = pack2_bf16(v0[rr].x, v1[rr].x); d.y = pack2_bf16(v0[rr].y, v1[rr].y);
            d.z = pack2_bf16(v0[rr].z, v1[rr].z); d.w = pack2_bf16(v0[rr].w, v1[rr].w);
            *reinterpret_cast<uint4*>(&xt[(static_cast<long>(tr0 + rr) * P2 + m) * kDC + l4 * 4]) = d;
          }
        }
      }
      const int du = D2 * (kDC / 4);
      for (int r0 = 0; r0 < TH; r0 += 4) {
        for (int i = tid; i < du; i += nthr) {
          const int m = i / (kDC / 4), l4 = i - m * (kDC / 4);
          const int w0 = 2 * m, w1 = w0 + 1;
          float4 v0[4], v1[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int hh = h0 + r0 + rr;
            const bool row_in = hh < H && r0 + rr < TH;
            const TD* dr = dy + ((n * H + (row_in ? hh : 0)) * static_cast<long>(W)) * C + cbase + l4 * 4;
            v0[rr] = row_in ? load4(dr + static_cast<long>(w0) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
            v1[rr] = (row_in && w1 < W) ? load4(dr + static_cast<long>(w1) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (r0 + rr >= TH) continue;
            uint4 d;
            d.x = pack2_bf16(v0[rr].x, v1[rr].x); d.y = pack2_bf16(v0[rr].y, v1[rr].y);
            d.z = pack2_bf16(v0[rr].z, v1[rr].z); d.w = pack2_bf16(v0[rr].w, v1[rr].w);
            *reinterpret_cast<uint4*>(&dt[(static_cast<long>(r0 + rr) * D2 + m) * kDC + l4 * 4]) = d;
          }
        }
      }
      __syncthreads();
      if (kh < 7) {
        for (int r = 0; r < TH; ++r) {
          const uint32_t* xrow = xt + static_cast<long>(r + kh) * P2 * kDC + lc;
          const uint32_t* drow = dt + static_cast<long>(r) * D2 * kDC + lc;
          uint32_t e0 = xrow[0], e1 = xrow[kDC], e2 = xrow[2 * kDC], e3 = xrow[3 * kDC];
          for (int m = 0; m < D2; ++m) {
            const uint32_t d = drow[m * kDC];
            const uint32_t o0 = __builtin_amdgcn_alignbit(e1, e0, 16), o1 = __builtin_amdgcn_alignbit(e2, e1, 16),
                           o2 = __builtin_amdgcn_alignbit(e3, e2, 16);
            acc[0] = dot2(d, e0, acc[0]); acc[1] = dot2(d, o0, acc[1]); acc[2] = dot2(d, e1, acc[2]);
            acc[3] = dot2(d, o1, acc[3]); acc[4] = dot2(d, e2, acc[4]); acc[5] = dot2(d, o2, acc[5]);
            acc[6] = dot2(d, e3, acc[6]);
            if (kh == 3) accb = dot2(d, ones, accb);
            e0 = e1; e1 = e2; e2 = e3;
            e3 = (m + 4 < P2) ? xrow[(m + 4) * kDC] : 0u;
          }
        }
      }
      __syncthreads();
    }
  }
  if (kh < 7) {
    float* pw = ws + static_cast<long>(blockIdx.x) * 50 * C + cbase + lc;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) pw[(kh * 7 + kw) * C] = acc[kw];
    if (kh == 3) pw[49 * C] = accb;
  }
}

// ------------------------------------------------------------------------------------------------
// Rolling-window variant of the filter gradient (same pipeline as dwconv7x7_roll_kernel): a workgroup walks down a band
// of `imgs` images x 32 channels with the 10-row x window as a ring in LDS and 4 dy rows per iteration; the rows of the
// next iteration are loaded into registers before the dot products of the current one run.  Every x row is staged once
// per band instead of 2.5 times, the global latency hides behind the arithmetic, and the 49 x 32 partial sums stay in
// registers over the whole band.  ws[part][50][C], part = blockIdx / (C/32).
// ------------------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void dwconv7x7_wgrad_roll_kernel(const TX* __restrict__ x, const uint16_t* __restrict__ dy,
                                                                   float* __restrict__ ws, long N, int H, int W, int C,
                                                                   int RS, int n_seg, int imgs) {
  constexpr int TH = 4, WR = TH + 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t wg_lds[];
  using Raw = typename std::conditional<sizeof(TX) == 4, float4, uint2>::type;
  const int D2 = (W + 1) / 2, P2 = D2 + 3;
  uint32_t* win = wg_lds;                                        // [WR][P2][32]
  uint32_t* dt = wg_lds + static_cast<long>(WR) * P2 * kDC;      // [TH][D2][32]
  const int n_cg = C / kDC;
  const int cbase = static_cast<int>(blockIdx.x % n_cg) * kDC;
  const long part = blockIdx.x / n_cg;
  const int seg = static_cast<int>(part % n_seg);
  const long n_begin = (part / n_seg) * imgs;
  const long n_end = n_begin + imgs < N ? n_begin + imgs : N;
  const int r_begin = seg * RS, r_end = min(H, r_begin + RS);
  const int tid = threadIdx.x;
  const int lc = tid & (kDC - 1), kh = tid / kDC;                // kh == 7: staging helper only

  // staging units (one per thread and row): x pair m covers padded columns 2m-3, 2m-2; dy pair m covers 2m, 2m+1
  const int xu = P2 * (kDC / 4), du = D2 * (kDC / 4);
  const bool xok = tid < xu, dok = tid < du;
  const int xi = min(tid, xu - 1), di = min(tid, du - 1);
  const int xm = xi / (kDC / 4), xl4 = xi % (kDC / 4), dm = di / (kDC / 4), dl4 = di % (kDC / 4);
  const int xw0 = 2 * xm - 3, xw1 = xw0 + 1, dw0 = 2 * dm, dw1 = dw0 + 1;
  const uint32_t xmask = ((xw0 >= 0 && xw0 < W) ? 0x0000ffffu : 0u) | ((xw1 >= 0 && xw1 < W) ? 0xffff0000u : 0u);
  const uint32_t dmask = 0x0000ffffu | (dw1 < W ? 0xffff0000u : 0u);
  const long xoff0 = static_cast<long>(min(max(xw0, 0), W - 1)) * C + cbase + xl4 * 4;
  const long xoff1 = static_cast<long>(min(max(xw1, 0), W - 1)) * C + cbase + xl4 * 4;
  const long doff0 = static_cast<long>(dw0) * C + cbase + dl4 * 4;
  const long doff1 = static_cast<long>(min(dw1, W - 1)) * C + cbase + dl4 * 4;

  auto load_x = [&](long n, int hh, Raw& v0, Raw& v1) {
    const TX* xr = x + ((n * H + min(max(hh, 0), H - 1)) * static_cast<long>(W)) * C;
    v0 = *reinterpret_cast<const Raw*>(xr + xoff0);
    v1 = *reinterpret_cast<const Raw*>(xr + xoff1);
  };
  auto store_x = [&](int slot, int hh, const Raw& v0, const Raw& v1) {
    if (!xok) return;
    uint4 d;
    if constexpr (sizeof(TX) == 4) {
      d.x = pack2_bf16(v0.x, v1.x); d.y = pack2_bf16(v0.y, v1.y); d.z = pack2_bf16(v0.z, v1.z); d.w = pack2_bf16(v0.w, v1.w);
    } else {
      d.x = (v0.x & 0xffffu) | (v1.x << 16); d.y = (v0.x >> 16) | (v1.x & 0xffff0000u);
      d.z = (v0.y & 0xffffu) | (v1.y << 16); d.w = (v0.y >> 16) | (v1.y & 0xffff0000u);
    }
    const uint32_t km = (hh >= 0 && hh < H) ? xmask : 0u;
    d.x &= km; d.y &= km; d.z &= km; d.w &= km;
    *reinterpret_cast<uint4*>(&win[(static_cast<long>(slot) * P2 + xm) * kDC + xl4 * 4]) = d;
  };
  auto load_d = [&](long n, int hh, uint2& v0, uint2& v1) {
    const uint16_t* dr = dy + ((n * H + min(hh, H - 1)) * static_cast<long>(W)) * C;
    v0 = *reinterpret_cast<const uint2*>(dr + doff0);
    v1 = *reinterpret_cast<const uint2*>(dr + doff1);
  };
  auto store_d = [&](int r, int hh, const uint2& v0, const uint2& v1) {
    if (!dok) return;
    uint4 d;
    d.x = (v0.x & 0xffffu) | (v1.x << 16); d.y = (v0.x >> 16) | (v1.x & 0xffff0000u);
    d.z = (v0.y & 0xffffu) | (v1.y << 16); d.w = (v0.y >> 16) | (v1.y & 0xffff0000u);
    const uint32_t km = hh < r_end ? dmask : 0u;                 // rows past the band contribute nothing
    d.x &= km; d.y &= km; d.z &= km; d.w &= km;
    *reinterpret_cast<uint4*>(&dt[(static_cast<long>(r) * D2 + dm) * kDC + dl4 * 4]) = d;
  };

  float acc[7], accb = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = 0.f;
  const uint32_t ones = 0x3f803f80u;                             // (1.0bf16, 1.0bf16)

  for (long n = n_begin; n < n_end; ++n) {
    // ---- fill: x rows r_begin-3 .. r_begin+6 -> slots 0..9, dy rows r_begin .. r_begin+3
    {
      Raw a0[5], a1[5];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int rr = 0; rr < 5; ++rr) load_x(n, r_begin - 3 + half * 5 + rr, a0[rr], a1[rr]);
#pragma unroll
        for (int rr = 0; rr < 5; ++rr) store_x(half * 5 + rr, r_begin - 3 + half * 5 + rr, a0[rr], a1[rr]);
      }
      uint2 b0[TH], b1[TH];
#pragma unroll
      for (int rr = 0; rr < TH; ++rr) load_d(n, r_begin + rr, b0[rr], b1[rr]);
#pragma unroll
      for (int rr = 0; rr < TH; ++rr) store_d(rr, r_begin + rr, b0[rr], b1[rr]);
    }
    __syncthreads();
    int s0 = 0;
    for (int h0 = r_begin; h0 < r_end; h0 += TH) {
      const bool more = h0 + TH < r_end;
      Raw px0[TH], px1[TH];
      uint2 pd0[TH], pd1[TH];
      if (more) {
#pragma unroll
        for (int rr = 0; rr < TH; ++rr) {
          load_x(n, h0 + TH + 3 + rr, px0[rr], px1[rr]);
          load_d(n, h0 + TH + rr, pd0[rr], pd1[rr]);
        }
      }
      if (kh < 7) {
#pragma unroll
        for (int r = 0; r < TH; ++r) {
          int slot = s0 + r + kh;
          if (slot >= WR) slot -= WR;
          if (slot >= WR) slot -= WR;
          const uint32_t* xrow = win + static_cast<long>(slot) * P2 * kDC + lc;
          const uint32_t* drow = dt + static_cast<long>(r) * D2 * kDC + lc;
          uint32_t e0 = xrow[0], e1 = xrow[kDC], e2 = xrow[2 * kDC], e3 = xrow[3 * kDC];
#pragma unroll 2
          for (int m = 0; m < D2; ++m) {
            const uint32_t d = drow[m * kDC];
            const uint32_t e4 = xrow[(m + 4) * kDC];             // m + 4 <= D2 + 3 = P2: one dword past the row at most (stays inside the ring / dy tile)
            const uint32_t o0 = __builtin_amdgcn_alignbit(e1, e0, 16), o1 = __builtin_amdgcn_alignbit(e2, e1, 16),
                           o2 = __builtin_amdgcn_alignbit(e3, e2, 16);
            acc[0] = dot2(d, e0, acc[0]); acc[1] = dot2(d, o0, acc[1]); acc[2] = dot2(d, e1, acc[2]);
            acc[3] = dot2(d, o1, acc[3]); acc[4] = dot2(d, e2, acc[4]); acc[5] = dot2(d, o2, acc[5]);
            acc[6] = dot2(d, e3, acc[6]);
            if (kh == 3) accb = dot2(d, ones, accb);
            e0 = e1; e1 = e2; e2 = e3; e3 = e4;
          }
        }
      }
      __syncthreads();
      if (more) {
#pragma unroll
        for (int rr = 0; rr < TH; ++rr) {
          int slot = s0 + rr;
          if (slot >= WR) slot -= WR;
          store_x(slot, h0 + TH + 3 + rr, px0[rr], px1[rr]);
          store_d(rr, h0 + TH + rr, pd0[rr], pd1[rr]);
        }
      }
      s0 += TH;
      if (s0 >= WR) s0 -= WR;
      __syncthreads();
    }
  }
  if (kh < 7) {
    float* pw = ws + part * 50 * C + cbase + lc;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) pw[(kh * 7 + kw) * C] = acc[kw];
    if (kh == 3) pw[49 * C] = accb;
  }
}

// out[j] = sum_p ws[p*len + j]: 16 outputs x 16 part-groups per block; every thread sums its parts (8 loads in flight)
// in a fixed order, the 16 group sums are combined by a fixed tree (deterministic).  `len` is a few hundred to a few
// thousand, `nparts` up to 1024: the old 64-outputs-per-block shape left a dozen blocks walking 256 parts serially.
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ ws, float* __restrict__ out0,
                                                           float* __restrict__ out1, int split, int len, int nparts) {
  __shared__ float part[16][17];
  const int jl = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + jl;
  float s = 0.f;
  if (j < len) {
    int p = pg;
    for (; p + 7 * 16 < nparts; p += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[static_cast<long>(p + u * 16) * len + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nparts; p += 16) s += ws[static_cast<long>(p) * len + j];
  }
  part[pg][jl] = s;
  __syncthreads();
  if (pg == 0 && j < len) {
    float t[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = part[g][jl];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int g = 0; g < w; ++g) t[g] += t[g + w];
    if (j < split) out0[j] = t[0]; else if (out1) out1[j - split] = t[0];
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over C of [M, C] rows (+ optional exact GELU).  GROUP lanes cooperate on one row,
// 64/GROUP rows per wavefront, 4 channels per lane-chunk, up to NV chunks per lane in registers.
// ------------------------------------------------------------------------------------------------
// GELU(z) = z * Phi(z) with erfc(|z|/sqrt 2) = 2^Q(|z|), Q of degree 5 (max |error| of GELU 1.2e-6; tools/fit_gelu.py;
// the same evaluation as the fused block kernels): one v_exp_f32, no division, no branch - libdevice's erff costs ~4x.
__device__ __forceinline__ float erfc_q(float az) {
  float q = fmaf(-0.00041175442346105595f, az, 0.006678475199902348f);
  q = fmaf(q, az, -0.050879760394516485f);
  q = fmaf(q, az, -0.46094072908550926f);
  q = fmaf(q, az, -1.150400682855232f);
  q = fmaf(q, az, -8.454223479528131e-05f);
  return __builtin_amdgcn_exp2f(q);
}
__device__ __forceinline__ float gelu_f(float z) {
  const float az = fabsf(z);
  return fmaf(az * erfc_q(az), -0.5f, fmaxf(z, 0.0f));
}
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float az = fabsf(z);
  const float he = 0.5f * erfc_q(az);
  const float cdf = z > 0.0f ? 1.0f - he : he;
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * z * z);
  return fmaf(z, pdf, cdf);
}
// out[j] = sum_s parts[s][j] for bf16 partial products of a split-K batched GEMM (fp32 accumulation, fixed order):
// block = 64 column chunks (8 columns = 16 bytes per lane) x 4 part groups; a thread walks parts pg, pg+4, ... with four
// loads in flight, the 4 group sums are combined through LDS.  (ATen's sum over dim 0 read the 75 MB of a 64 x 1536 x 384
// stack at 2.1 TB/s.)
__global__ __launch_bounds__(256) void sum_parts_bf16_kernel(const uint16_t* __restrict__ parts, float* __restrict__ out,
                                                             long S, long L) {
  __shared__ float red[4][64][9];
  const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const long j = (static_cast<long>(blockIdx.x) * 64 + cl) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (j < L) {
    long sidx = pg;
    for (; sidx + 12 < S; sidx += 16) {
      uint4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const uint4*>(parts + (sidx + 4 * u) * L + j);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t w[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w[e] << 16); acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
    }
    for (; sidx < S; sidx += 4) {
      const uint4 t = *reinterpret_cast<const uint4*>(parts + sidx * L + j);
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w[e] << 16); acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[pg][cl][e] = acc[e];
  __syncthreads();
  if (pg == 0 && j < L) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((red[0][cl][e] + red[1][cl][e]) + red[2][cl][e]) + red[3][cl][e];
    *reinterpret_cast<float4*>(out + j) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(out + j + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = GROUP / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

template <typename TX, typename TY, int GROUP, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float eps,
                                                            TY* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, long M, int C, int gelu) {
  constexpr int RPB = 256 / GROUP;                    // rows per block per iteration
  const int gl = threadIdx.x % GROUP, gr = threadIdx.x / GROUP;
  const float invC = 1.0f / static_cast<float>(C);
  for (long row = static_cast<long>(blockIdx.x) * RPB + gr; row < M; row += static_cast<long>(gridDim.x) * RPB) {
    const TX* xr = x + row * C;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (gl + k * GROUP) * 4;
      if (c < C) { v[k] = load4(xr + c); s += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
    }
    const float mu = group_sum<GROUP>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (gl + k * GROUP) * 4;
      if (c < C) {
        const float a = v[k].x - mu, b = v[k].y - mu, cc = v[k].z - mu, d = v[k].w - mu;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rs = rsqrtf(group_sum<GROUP>(q) * invC + eps);
    if (gl == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    TY* yr = y + row * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (gl + k * GROUP) * 4;
      if (c < C) {
        const float4 w4 = load4(weight + c), b4 = load4(bias + c);
        float4 o;
        o.x = (v[k].x - mu) * rs * w4.x + b4.x; o.y = (v[k].y - mu) * rs * w4.y + b4.y;
        o.z = (v[k].z - mu) * rs * w4.z + b4.z; o.w = (v[k].w - mu) * rs * w4.w + b4.w;
        if (gelu) { o.x = gelu_f(o.x); o.y = gelu_f(o.y); o.z = gelu_f(o.z); o.w = gelu_f(o.w); }
        store4(yr + c, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Element-wise tails of the library-path MLP (widths without a fused block kernel), each ONE pass over its operands with the
// per-channel parameter gradients accumulated on the way (deterministic partials -> reduce_parts_kernel):
//   scale_residual_kernel      out = x + gamma * y                                   (models/convnext.py:47-49)
//   scale_residual_bwd_kernel  dO = bf16(g * gamma);  dgamma = sum_m g*y;  db2 = sum_m dO
//   gelu_bwd_colsum_kernel     dHpre = bf16(dH * GELU'(Hpre));  db1 = sum_m dHpre
// Thread = 4 consecutive channels; blockIdx.x walks column tiles of 1024, blockIdx.y strides over rows.
// ------------------------------------------------------------------------------------------------
// row lanes (= partial sums per column) of the column-sum passes: enough threads to fill the chip also for narrow tensors
// (C = 96: 24 threads per row lane; 2048 lanes were 192 blocks and ran the 462 MB d(gamma)/d(b2) pass at 1.1 TB/s)
inline int col_parts(int n_cols) {
  const int p = (1 << 19) / n_cols;
  return p < 2048 ? 2048 : (p > 8192 ? 8192 : p);
}
__device__ __forceinline__ float round_bf16(float v) { return __uint_as_float(static_cast<uint32_t>(f2bf(v)) << 16); }
__device__ __forceinline__ float4 round_bf16(float4 v) { return make_float4(round_bf16(v.x), round_bf16(v.y), round_bf16(v.z), round_bf16(v.w)); }

// thread layout shared by the three kernels: a block is CQ channel-quads x RY rows (CQ = min(C/4, 256), RY = 256 / CQ), so
// narrow tensors (C = 96: 24 quads) still fill their blocks; "row lane" rl = blockIdx.y*RY + ty walks rows rl, rl+R, ...
struct ColMap { int c, rl, R; bool active; };
// CQ = the divisor of C/4 (<= 256) that keeps most of the 256 threads busy (C/4 = 384 -> 128 x 2 rows, 24 -> 24 x 10 rows)
__host__ __device__ inline int col_cq(int cq_all) {
  int best = 1, best_use = 0;
  for (int d = 1; d <= 256 && d <= cq_all; ++d) {
    if (cq_all % d) continue;
    const int use = d * (256 / d);
    if (use >= best_use) { best_use = use; best = d; }
  }
  return best;
}
__device__ __forceinline__ ColMap col_map(int C, int CQ) {
  const int RY = 256 / CQ;
  const int tx = threadIdx.x % CQ, ty = threadIdx.x / CQ;
  ColMap m;
  m.c = (blockIdx.x * CQ + tx) * 4;
  m.rl = blockIdx.y * RY + ty;
  m.R = gridDim.y * RY;
  m.active = ty < RY && m.c < C;
  return m;
}
inline dim3 col_grid(int C, long M, long max_lanes, int* cq_out) {
  const int cq_all = C / 4, CQ = col_cq(cq_all), RY = 256 / CQ;
  *cq_out = CQ;
  long lanes = M < max_lanes ? M : max_lanes;
  long gy = (lanes + RY - 1) / RY;
  if (gy * RY > max_lanes) gy = max_lanes / RY;
  if (gy < 1) gy = 1;
  return dim3(cq_all / CQ, static_cast<unsigned>(gy));
}
inline int col_lanes(int CQ, const dim3& grid) { return static_cast<int>(grid.y) * (256 / CQ); }

template <typename TX, typename TO>
__global__ __launch_bounds__(256) void scale_residual_kernel(const TX* __restrict__ x, const uint16_t* __restrict__ y,
                                                             const float* __restrict__ gamma, TO* __restrict__ out, long M, int C,
                                                             int CQ) {
  const ColMap cm = col_map(C, CQ);
  if (!cm.active) return;
  const int c = cm.c;
  const float4 g4 = gamma ? load4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  for (long m = cm.rl; m < M; m += cm.R) {
    const float4 xv = load4(x + m * C + c), yv = load4(y + m * C + c);
    store4(out + m * C + c, make_float4(fmaf(yv.x, g4.x, xv.x), fmaf(yv.y, g4.y, xv.y), fmaf(yv.z, g4.z, xv.z), fmaf(yv.w, g4.w, xv.w)));
  }
}

template <typename TG>
__global__ __launch_bounds__(256) void scale_residual_bwd_kernel(const TG* __restrict__ g, const uint16_t* __restrict__ y,
                                                                 const float* __restrict__ gamma, uint16_t* __restrict__ dos,
                                                                 float* __restrict__ ws, long M, int C, int CQ) {
  const ColMap cm = col_map(C, CQ);
  if (!cm.active) return;
  const int c = cm.c;
  const float4 g4 = gamma ? load4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  // 4 rows per trip: the loads of a trip are independent (one load in flight per thread left this pass latency-bound at
  // ~2 TB/s on cache-resident tensors); the sums keep their row order
  constexpr int UR = 4;
  for (long m0 = cm.rl; m0 < M; m0 += static_cast<long>(UR) * cm.R) {
    float4 gv[UR], yv[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) {
      const long m = m0 + static_cast<long>(k) * cm.R;
      const long mc = m < M ? m : M - 1;
      gv[k] = load4(g + mc * C + c);
      if (ws && y) yv[k] = load4(y + mc * C + c);
    }
#pragma unroll
    for (int k = 0; k < UR; ++k) {
      const long m = m0 + static_cast<long>(k) * cm.R;
      if (m >= M) break;
      const float4 d = round_bf16(make_float4(gv[k].x * g4.x, gv[k].y * g4.y, gv[k].z * g4.z, gv[k].w * g4.w));
      if (dos) store4(dos + m * C + c, d);
      if (ws) {                                                    // sums of the rounded values, as summing the bf16 tensor would
        ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
        if (y) { ag.x = fmaf(gv[k].x, yv[k].x, ag.x); ag.y = fmaf(gv[k].y, yv[k].y, ag.y); ag.z = fmaf(gv[k].z, yv[k].z, ag.z); ag.w = fmaf(gv[k].w, yv[k].w, ag.w); }
      }
    }
  }
  if (ws) {
    float* p = ws + static_cast<long>(cm.rl) * 2 * C;
    store4(p + c, ag);
    store4(p + C + c, ab);
  }
}

// d(gamma) of a block WITHOUT a pass over g and y2 (round 5).  With y2 = H W2^T + b2 (models/convnext.py:44-47):
//     d(gamma)[c] = sum_m g[m,c] y2[m,c] = sum_j W2[c,j] (g^T H)[c,j] + b2[c] sum_m g[m,c],
// and the weight gradients the training pass has just computed are dW2 = dO^T H and d(b2) = sum_m dO with dO = bf16(g gamma), so
//     d(gamma)[c] = (sum_j W2[c,j] dW2[c,j] + b2[c] d(b2)[c]) / gamma[c]        (W2 rounded to bf16 as the forward's GEMM read it)
// - a reduction over the [C, 4C] matrices instead of 462 MB of g and y2 at 56 x 56 x 96.  Differs from the direct sum by the bf16 rounding
// of dO and of y2 (relative L2 2e-3 of d(gamma), measured; the direct sum's own rounding of y2: 1.4e-3).  One block per channel; a
// channel whose gamma is exactly zero has no dO to recover g from and takes the direct sum over its column of g and y2 - y2 as stored, or,
// when the forward did not keep it, recomputed from the H tiles (strided, slow, exact; a gamma lands on 0.0f with probability ~2^-24 per
// sign change).
template <typename TG>
__global__ __launch_bounds__(256) void block_dgamma_kernel(const float* __restrict__ w2, const float* __restrict__ dw2,
                                                           const float* __restrict__ b2, const float* __restrict__ db2,
                                                           const float* __restrict__ gamma, const TG* __restrict__ g,
                                                           const uint16_t* __restrict__ y2, const uint16_t* __restrict__ ht,
                                                           float* __restrict__ dgamma, long M, int C, int Hd) {
  __shared__ float red[256];
  const int c = blockIdx.x, t = threadIdx.x;
  const float gm = gamma[c];
  // the identity divides by gamma what dO = bf16(g gamma) carried: sound while that product is a normal bf16 number, i.e. for every
  // gamma a model can hold (the layer-scale init 1e-6 included: tests/test_gpu_model_ops.py); below 1e-30 (and at 0 or NaN) dO has
  // flushed and the channel takes the direct sum
  const bool ident = fabsf(gm) >= 1e-30f;
  float s = 0.f;
  if (ident) {
    for (int j = t; j < Hd; j += 256) s = fmaf(round_bf16(w2[static_cast<long>(c) * Hd + j]), dw2[static_cast<long>(c) * Hd + j], s);
  } else {
    for (long m = t; m < M; m += 256) {
      float gv, yv;
      if constexpr (sizeof(TG) == 4) gv = g[m * C + c]; else gv = __uint_as_float(static_cast<uint32_t>(g[m * C + c]) << 16);
      if (y2) {
        yv = __uint_as_float(static_cast<uint32_t>(y2[m * C + c]) << 16);
      } else {
        // y2[m, c] recomputed from the H tiles (CNX_TN_ACC: tiles [M/32][Hd/32] of 2 KiB, element (r, n) of a tile at byte
        // 64 r + 32 ((n/4) % 2) + 8 (n/8) + 2 (n % 4)), rounded to bf16 as the forward stored it
        const uint16_t* tile_row = ht + (m / 32) * (static_cast<long>(Hd) / 32) * 1024 + (m % 32) * 32;
        float acc = 0.f;
        for (int j = 0; j < Hd; ++j) {
          const int n = j % 32;
          const uint16_t hv = tile_row[static_cast<long>(j / 32) * 1024 + 16 * ((n / 4) % 2) + 4 * (n / 8) + (n % 4)];
          acc = fmaf(__uint_as_float(static_cast<uint32_t>(hv) << 16), round_bf16(w2[static_cast<long>(c) * Hd + j]), acc);
        }
        yv = round_bf16(acc + (b2 ? b2[c] : 0.f));
      }
      s = fmaf(gv, yv, s);
    }
  }
  red[t] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {                  // fixed tree: deterministic
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t == 0) dgamma[c] = ident ? (red[0] + (b2 ? b2[c] * db2[c] : 0.f)) / gm : red[0];
}

// The LayerNorm parameter gradients of a block the same way (round 5).  With a = LN(u) = xh ln_w + ln_b (bf16, the left operand of the
// first linear layer), da = dHpre W1 and the weight gradients dW1 = dHpre^T a, d(b1) = sum_m dHpre:
//     d(ln_b)[c] = sum_m da[m,c]           = sum_j W1[j,c] d(b1)[j]
//     d(ln_w)[c] = sum_m da[m,c] xh[m,c]   = (sum_j W1[j,c] dW1[j,c] - ln_b[c] d(ln_b)[c]) / ln_w[c]
// (W1 rounded to bf16 as the GEMMs read it).  Conditioning (round 6): the second identity recovers xh from a = bf16(xh ln_w + ln_b), whose
// rounding error 2^-9 |a| comes back divided by ln_w - a relative error of ~2^-9 max(1, |ln_b| / |ln_w|) on the channel, where the
// reference's fp32 LayerNorm backward has none.  A channel with |ln_b| > 4 |ln_w| (or ln_w == 0, or a NaN) is ILL-CONDITIONED for the
// identity and takes the direct sum over da, u, mean, rstd: dln_ill() is the one predicate both kernels below evaluate.
//   block_dln_direct_kernel (wide grid, launched first when the caller passes a workspace): every workgroup derives the list of
//       ill-conditioned channels from ln_w / ln_b and leaves at once when it is empty (the benchmarked model at its init and every
//       LayerNorm with |ln_b| <= 4 |ln_w|: ~2 us); otherwise it sums da xh over its rows for the listed channels - da as stored, or
//       recomputed row by row from the dHpre tiles and the listed columns of W1 (staged in LDS) - and writes ws[wg][c];
//   block_dln_kernel: the identities; for a listed channel the fixed-order sum of the partials (without a workspace: its own serial
//       direct sum, correct and slow - the pre-round-6 path for an exactly-zero ln_w).
__device__ __forceinline__ bool dln_ill(float lw, float lb) { return !(fabsf(lw) * 4.0f >= fabsf(lb)) || lw == 0.f; }

constexpr int kDlnParts = 256, kDlnChunk = 8;

__global__ __launch_bounds__(256) void block_dln_direct_kernel(const float* __restrict__ w1, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, const uint16_t* __restrict__ da,
                                                                const uint16_t* __restrict__ dhpt, const uint16_t* __restrict__ u,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ ws, long M, int C, int Hd) {
  extern __shared__ float w1s[];                           // [Hd][kDlnChunk]: bf16-rounded W1 columns of the chunk's channels (tile form)
  __shared__ int list[2048];
  __shared__ int nlist;
  __shared__ float red[256];
  const int t = threadIdx.x;
  if (t < 64) {                                            // ordered compaction of the ill-conditioned channels by the first wavefront
    int base = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + t;
      const bool ill = c < C && dln_ill(ln_w[c < C ? c : 0], ln_b[c < C ? c : 0]);
      const unsigned long long mk = __ballot(ill);
      if (ill) list[base + __popcll(mk & ((1ull << t) - 1ull))] = c;
      base += __popcll(mk);
    }
    if (t == 0) nlist = base;
  }
  __syncthreads();
  const int nf = nlist;
  if (nf == 0) return;
  // rows of this workgroup: whole 32-row tiles
  const long tiles = (M + 31) / 32, per = (tiles + gridDim.x - 1) / gridDim.x;
  const long m_lo = static_cast<long>(blockIdx.x) * per * 32;
  long m_hi = m_lo + per * 32;
  if (m_hi > M) m_hi = M;
  for (int f0 = 0; f0 < nf; f0 += kDlnChunk) {
    const int nfc = nf - f0 < kDlnChunk ? nf - f0 : kDlnChunk;
    if (!da) {
      __syncthreads();
      for (int i = t; i < Hd * kDlnChunk; i += 256) {
        const int j = i / kDlnChunk, f = i % kDlnChunk;
        w1s[i] = f < nfc ? round_bf16(w1[static_cast<long>(j) * C + list[f0 + f]]) : 0.f;
      }
      __syncthreads();
    }
    float acc[kDlnChunk];
#pragma unroll
    for (int f = 0; f < kDlnChunk; ++f) acc[f] = 0.f;
    for (long m = m_lo + t; m < m_hi; m += 256) {
      float dav[kDlnChunk];
#pragma unroll
      for (int f = 0; f < kDlnChunk; ++f) dav[f] = 0.f;
      if (da) {
#pragma unroll
        for (int f = 0; f < kDlnChunk; ++f)
          if (f < nfc) dav[f] = __uint_as_float(static_cast<uint32_t>(da[m * C + list[f0 + f]]) << 16);
      } else {
        // row m of the dHpre tiles (CNX_TN_ACC): 64 contiguous bytes per hidden block, value p of them is hidden unit
        // 8 ((p / 4) % 4) + 4 (p / 16) + p % 4 of the block (the inverse of byte 64 r + 32 ((n/4) % 2) + 8 (n/8) + 2 (n % 4))
        const uint4* trow = reinterpret_cast<const uint4*>(dhpt + (m / 32) * (static_cast<long>(Hd) / 32) * 1024 + (m % 32) * 32);
        for (int jb = 0; jb < Hd / 32; ++jb) {
          uint4 q[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = trow[static_cast<long>(jb) * 128 + k];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t wd[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int pidx = 8 * k + e;
              const int n = 8 * ((pidx / 4) % 4) + 4 * (pidx / 16) + pidx % 4;
              const float dv = (e & 1) ? __uint_as_float(wd[e / 2] & 0xffff0000u) : __uint_as_float(wd[e / 2] << 16);
              const float* wr = w1s + (jb * 32 + n) * kDlnChunk;
#pragma unroll
              for (int f = 0; f < kDlnChunk; ++f) dav[f] = fmaf(dv, wr[f], dav[f]);
            }
          }
        }
      }
      const float mu = mean[m], rs = rstd[m];
#pragma unroll
      for (int f = 0; f < kDlnChunk; ++f)
        if (f < nfc) {
          const float xh = (__uint_as_float(static_cast<uint32_t>(u[m * C + list[f0 + f]]) << 16) - mu) * rs;
          acc[f] = fmaf(dav[f], xh, acc[f]);
        }
    }
    for (int f = 0; f < nfc; ++f) {                        // fixed tree per channel: deterministic
      __syncthreads();
      red[t] = acc[f];
      __syncthreads();
#pragma unroll
      for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
      }
      if (t == 0) ws[static_cast<long>(blockIdx.x) * C + list[f0 + f]] = red[0];
    }
  }
}

__global__ __launch_bounds__(1024) void block_dln_kernel(const float* __restrict__ w1, const float* __restrict__ dw1,
                                                         const float* __restrict__ db1, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, const uint16_t* __restrict__ da,
                                                         const uint16_t* __restrict__ dhpt, const uint16_t* __restrict__ u, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dlw, float* __restrict__ dlb,
                                                         const float* __restrict__ ws, int parts, long M, int C, int Hd) {
  // a block = 32 channels x 32 row lanes of [Hd, C]: every load is a 128-byte run of a row, four rows per lane in flight (a launch is
  // a handful of blocks: what it costs is the length of a lane's chain of load latencies - 8 row lanes and one row at a time ran
  // 117 us, one block per channel with its column read at a stride of C floats 10 us)
  constexpr int RL = 32;
  __shared__ float red[2][RL][32];
  __shared__ float dsum[1024];
  __shared__ unsigned zmask;
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int c = blockIdx.x * 32 + tx;
  float sw = 0.f, sb = 0.f;
  if (c < C) {
    int j = ty;
    for (; j + 3 * RL < Hd; j += 4 * RL) {
      float w[4], d[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w[k] = w1[static_cast<long>(j + k * RL) * C + c]; d[k] = dw1[static_cast<long>(j + k * RL) * C + c]; b[k] = db1[j + k * RL];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float wr = round_bf16(w[k]); sw = fmaf(wr, d[k], sw); sb = fmaf(wr, b[k], sb); }
    }
    for (; j < Hd; j += RL) {
      const float wr = round_bf16(w1[static_cast<long>(j) * C + c]);
      sw = fmaf(wr, dw1[static_cast<long>(j) * C + c], sw);
      sb = fmaf(wr, db1[j], sb);
    }
  }
  red[0][ty][tx] = sw; red[1][ty][tx] = sb;
  if (t < 64) {                                          // the ill-conditioned channels of this block (none in practice): one ballot finds them
    const unsigned long long zb = __ballot(c < C && dln_ill(ln_w[c < C ? c : 0], ln_b[c < C ? c : 0]));
    if (t == 0) zmask = static_cast<unsigned>(zb & 0xffffffffull);
  }
  __syncthreads();
  if (ty == 0) {
#pragma unroll
    for (int r = 1; r < RL; ++r) { sw += red[0][r][tx]; sb += red[1][r][tx]; }    // fixed order
    red[0][0][tx] = sw; red[1][0][tx] = sb;
  }
  __syncthreads();
  if (ws) {
    // the direct sums of block_dln_direct_kernel: `parts` partials per listed channel, summed in a fixed order (32 row lanes of
    // parts / 32 consecutive partials each, then the lanes in order)
    if (zmask != 0u) {
      float pd = 0.f;
      if (c < C && ((zmask >> tx) & 1u)) {
        const int per = (parts + RL - 1) / RL;
        for (int q = ty * per; q < (ty + 1) * per && q < parts; ++q) pd += ws[static_cast<long>(q) * C + c];
      }
      dsum[ty * 32 + tx] = pd;
      __syncthreads();
      if (ty == 0 && ((zmask >> tx) & 1u)) {
        float sdir = 0.f;
#pragma unroll
        for (int r = 0; r < RL; ++r) sdir += dsum[r * 32 + tx];
        red[0][0][tx] = sdir;
      }
      __syncthreads();
    }
  } else {
    for (unsigned mk = zmask; mk != 0u; mk &= mk - 1u) {   // no workspace: the serial direct sum, all threads per channel
      const int k = __builtin_ctz(mk);
      const int ck = blockIdx.x * 32 + k;
      float sd = 0.f;
      for (long m = t; m < M; m += 1024) {
        const float xh = (__uint_as_float(static_cast<uint32_t>(u[m * C + ck]) << 16) - mean[m]) * rstd[m];
        float dav;
        if (da) {
          dav = __uint_as_float(static_cast<uint32_t>(da[m * C + ck]) << 16);
        } else {                                           // da[m, c] = sum_j dHpre[m, j] W1[j, c] from the CNX_TN_ACC tiles (block_dgamma_kernel)
          const uint16_t* tile_row = dhpt + (m / 32) * (static_cast<long>(Hd) / 32) * 1024 + (m % 32) * 32;
          dav = 0.f;
          for (int j = 0; j < Hd; ++j) {
            const int n = j % 32;
            const uint16_t dv = tile_row[static_cast<long>(j / 32) * 1024 + 16 * ((n / 4) % 2) + 4 * (n / 8) + (n % 4)];
            dav = fmaf(__uint_as_float(static_cast<uint32_t>(dv) << 16), round_bf16(w1[static_cast<long>(j) * C + ck]), dav);
          }
        }
        sd = fmaf(dav, xh, sd);
      }
      dsum[t] = sd;
      __syncthreads();
      for (int w = 512; w > 0; w >>= 1) {
        if (t < w) dsum[t] += dsum[t + w];
        __syncthreads();
      }
      if (t == 0) red[0][0][k] = dsum[0];
      __syncthreads();
    }
  }
  if (ty == 0 && c < C) {
    const float lw = ln_w[c], b = red[1][0][tx], w = red[0][0][tx];
    dlb[c] = b;
    dlw[c] = !dln_ill(lw, ln_b[c]) ? (w - ln_b[c] * b) / lw : w;
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const uint16_t* __restrict__ dh, const uint16_t* __restrict__ hpre,
                                                              uint16_t* __restrict__ dhpre, float* __restrict__ ws, long M, int N,
                                                              int CQ) {
  const ColMap cm = col_map(N, CQ);
  if (!cm.active) return;
  const int c = cm.c;
  float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long m = cm.rl; m < M; m += cm.R) {
    const float4 d = load4(dh + m * N + c), z = load4(hpre + m * N + c);
    const float4 o = round_bf16(make_float4(d.x * gelu_grad_f(z.x), d.y * gelu_grad_f(z.y), d.z * gelu_grad_f(z.z), d.w * gelu_grad_f(z.w)));
    store4(dhpre + m * N + c, o);
    if (ws) { ab.x += o.x; ab.y += o.y; ab.z += o.z; ab.w += o.w; }
  }
  if (ws) store4(ws + static_cast<long>(cm.rl) * N + c, ab);
}

constexpr int kLnBwdBlocks = 1024;

template <typename TD, typename TX, typename TO, int GROUP, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TO* __restrict__ dx,
                                                            float* __restrict__ ws, long M, int C, int gelu,
                                                            const float* __restrict__ add) {
  constexpr int RPB = 256 / GROUP;
  __shared__ float red[2][RPB][GROUP * 4];            // per row-group partial parameter gradients (one chunk)
  const int gl = threadIdx.x % GROUP, gr = threadIdx.x / GROUP;
  const float invC = 1.0f / static_cast<float>(C);
  float4 w4[NV], b4[NV], aw[NV], ab[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (gl + k * GROUP) * 4;
    aw[k] = ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) { w4[k] = load4(weight + c); b4[k] = (gelu && bias) ? load4(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f); }
  }
  for (long row = static_cast<long>(blockIdx.x) * RPB + gr; row < M; row += static_cast<long>(gridDim.x) * RPB) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (gl + k * GROUP) * 4;
      if (c < C) {
        const float4 xv = load4(x + row * C + c);
        float4 d = load4(dy + row * C + c);
        xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        if (gelu) {
          d.x *= gelu_grad_f(xh[k].x * w4[k].x + b4[k].x); d.y *= gelu_grad_f(xh[k].y * w4[k].y + b4[k].y);
          d.z *= gelu_grad_f(xh[k].z * w4[k].z + b4[k].z); d.w *= gelu_grad_f(xh[k].w * w4[k].w + b4[k].w);
        }
        if (ws) {
          aw[k].x += d.x * xh[k].x; aw[k].y += d.y * xh[k].y; aw[k].z += d.z * xh[k].z; aw[k].w += d.w * xh[k].w;
          ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
        }
        g[k] = make_float4(d.x * w4[k].x, d.y * w4[k].y, d.z * w4[k].z, d.w * w4[k].w);
        s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
        s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
      }
    }
    const float c1 = group_sum<GROUP>(s1) * invC, c2 = group_sum<GROUP>(s2) * invC;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (gl + k * GROUP) * 4;
      if (c < C) {
        float4 o;
        o.x = rs * (g[k].x - c1 - xh[k].x * c2); o.y = rs * (g[k].y - c1 - xh[k].y * c2);
        o.z = rs * (g[k].z - c1 - xh[k].z * c2); o.w = rs * (g[k].w - c1 - xh[k].w * c2);
        if (add) { const float4 a4 = load4(add + row * C + c); o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w; }
        store4(dx + row * C + c, o);
      }
    }
  }
  if (!ws) return;
  // combine the RPB row-groups of this block (fixed order), one channel chunk at a time
  float* pw = ws + static_cast<long>(blockIdx.x) * 2 * C;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (gl + k * GROUP) * 4;
    __syncthreads();
    *reinterpret_cast<float4*>(&red[0][gr][gl * 4]) = aw[k];
    *reinterpret_cast<float4*>(&red[1][gr][gl * 4]) = ab[k];
    __syncthreads();
    if (gr == 0 && c < C) {
      float4 sw = make_float4(0.f, 0.f, 0.f, 0.f), sb = sw;
#pragma unroll
      for (int r = 0; r < RPB; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(&red[0][r][gl * 4]);
        const float4 b = *reinterpret_cast<const float4*>(&red[1][r][gl * 4]);
        sw.x += a.x; sw.y += a.y; sw.z += a.z; sw.w += a.w; sb.x += b.x; sb.y += b.y; sb.z += b.z; sb.w += b.w;
      }
      store4(pw + c, sw);
      store4(pw + C + c, sb);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide variants for the widths of the model (C = 48 * 2^k: 48 ... 768): G = C/24 lanes per row, three 8-channel chunks
// (16 bytes of bf16, 32 of fp32) per lane, 64/G rows per wavefront.  The 4-channel kernels above keep one row of <= 1.5 KB in
// flight per wavefront and two 6-step reductions per row: latency-bound at ~3 TB/s on tensors that sit in the MALL.
// ------------------------------------------------------------------------------------------------
struct F8 { float v[8]; };
__device__ __forceinline__ F8 load8(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  return F8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ F8 load8(const uint16_t* p) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  return F8{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u),
             __uint_as_float(t.z << 16), __uint_as_float(t.z & 0xffff0000u), __uint_as_float(t.w << 16), __uint_as_float(t.w & 0xffff0000u)}};
}
__device__ __forceinline__ void store8(float* p, const F8& o) {
  *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(o.v[4], o.v[5], o.v[6], o.v[7]);
}
__device__ __forceinline__ void store8(uint16_t* p, const F8& o) {
  uint4 t;
  t.x = static_cast<uint32_t>(f2bf(o.v[0])) | (static_cast<uint32_t>(f2bf(o.v[1])) << 16);
  t.y = static_cast<uint32_t>(f2bf(o.v[2])) | (static_cast<uint32_t>(f2bf(o.v[3])) << 16);
  t.z = static_cast<uint32_t>(f2bf(o.v[4])) | (static_cast<uint32_t>(f2bf(o.v[5])) << 16);
  t.w = static_cast<uint32_t>(f2bf(o.v[6])) | (static_cast<uint32_t>(f2bf(o.v[7])) << 16);
  *reinterpret_cast<uint4*>(p) = t;
}

// Element offset of row (n, h, w) of an [N, pH, pW, C] tensor in its 2x2-patch form [N, pH/2, pW/2, (h&1, w&1, C)] - the
// left operand of the downsample "convolution" (kernel 2, stride 2 = a GEMM over patches); pW == 0: plain rows.
__device__ __forceinline__ long patch2_offset(long row, int C, int pH, int pW) {
  if (pW == 0) return row * C;
  const int w = static_cast<int>(row % pW);
  const long t = row / pW;
  const int h = static_cast<int>(t % pH);
  const long n = t / pH;
  return (((n * (pH >> 1) + (h >> 1)) * (pW >> 1) + (w >> 1)) * 4 + (h & 1) * 2 + (w & 1)) * C;
}

template <typename TX, typename TY, int G>
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(const TX* __restrict__ x, const float* __restrict__ weight,
                                                                 const float* __restrict__ bias, float eps,
                                                                 TY* __restrict__ y, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, long M, int C, int gelu, int pH,
                                                                 int pW) {
  constexpr int RPB = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const float invC = 1.0f / static_cast<float>(C);
  F8 w8[3], b8[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { w8[k] = load8(weight + (gl + k * G) * 8); b8[k] = load8(bias + (gl + k * G) * 8); }
  for (long row = static_cast<long>(blockIdx.x) * RPB + gr; row < M; row += static_cast<long>(gridDim.x) * RPB) {
    const TX* xr = x + row * C;
    F8 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = load8(xr + (gl + k * G) * 8);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[k].v[e];
    const float mu = group_sum<G>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k].v[e] - mu; q += d * d; }
    const float rs = rsqrtf(group_sum<G>(q) * invC + eps);
    if (gl == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    TY* yr = y + patch2_offset(row, C, pH, pW);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      F8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o.v[e] = (v[k].v[e] - mu) * rs * w8[k].v[e] + b8[k].v[e];
        if (gelu) o.v[e] = gelu_f(o.v[e]);
      }
      store8(yr + (gl + k * G) * 8, o);
    }
  }
}

// SUMS = false (round 5): the input-gradient-only calls of the attack, and the training calls whose parameter gradients come from
// cnx_block_dln - without the 48 accumulator registers per lane the kernel fits more wavefronts per SIMD
template <typename TD, typename TX, typename TO, int G, bool SUMS = true>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, TO* __restrict__ dx,
                                                                 float* __restrict__ ws, long M, int C, int gelu, int pH,
                                                                 int pW, const float* __restrict__ add) {
  constexpr int RPB = 256 / G;
  __shared__ float red[SUMS ? 2 : 1][SUMS ? RPB : 1][SUMS ? G * 8 : 1];                // per row-group partial parameter gradients (one chunk)
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const float invC = 1.0f / static_cast<float>(C);
  F8 w8[3], b8[3], aw[SUMS ? 3 : 1], ab[SUMS ? 3 : 1];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    w8[k] = load8(weight + (gl + k * G) * 8);
    if (gelu && bias) b8[k] = load8(bias + (gl + k * G) * 8);
    else b8[k] = F8{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if constexpr (SUMS) aw[k] = ab[k] = F8{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  }
  for (long row = static_cast<long>(blockIdx.x) * RPB + gr; row < M; row += static_cast<long>(gridDim.x) * RPB) {
    const float mu = mean[row], rs = rstd[row];
    F8 xh[3], g[3];
    const TD* dyr = dy + patch2_offset(row, C, pH, pW);
#pragma unroll
    for (int k = 0; k < 3; ++k) { xh[k] = load8(x + row * C + (gl + k * G) * 8); g[k] = load8(dyr + (gl + k * G) * 8); }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h = (xh[k].v[e] - mu) * rs;
        float d = g[k].v[e];
        if (gelu) d *= gelu_grad_f(h * w8[k].v[e] + b8[k].v[e]);
        if constexpr (SUMS) { if (ws) { aw[k].v[e] += d * h; ab[k].v[e] += d; } }
        const float t = d * w8[k].v[e];
        xh[k].v[e] = h;
        g[k].v[e] = t;
        s1 += t;
        s2 += t * h;
      }
    const float c1 = group_sum<G>(s1) * invC, c2 = group_sum<G>(s2) * invC;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      F8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = rs * (g[k].v[e] - c1 - xh[k].v[e] * c2);
      if (add) {                                      // + the gradient that reached x along the skip connection (fp32, same shape)
        const F8 a8 = load8(add + row * C + (gl + k * G) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] += a8.v[e];
      }
      store8(dx + row * C + (gl + k * G) * 8, o);
    }
  }
  if constexpr (SUMS) {
  if (!ws) return;
  float* pw = ws + static_cast<long>(blockIdx.x) * 2 * C;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][gr][gl * 8 + e] = aw[k].v[e]; red[1][gr][gl * 8 + e] = ab[k].v[e]; }
    __syncthreads();
    // G*8 channels of this chunk x 2 sums, combined over the RPB row-groups in a fixed order
    for (int j = threadIdx.x; j < 2 * G * 8; j += 256) {
      const int which = j / (G * 8), cc = j % (G * 8);
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < RPB; ++r) t += red[which][r][cc];
      pw[which * C + k * G * 8 + cc] = t;
    }
  }
  }
}

// y = GELU(x) on bf16, exact-erf form (the activation between the two library GEMMs of the MLPs that have no fused block
// kernel: ConvNeXt C = 384 (training pass) / 768, ViT): 16 bytes per lane, two chunks in flight.  ATen's kernel ran the
// [50 432, 3072] tensor of ViT-B at 4.8 TB/s.
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long n8) {
  const long stride = static_cast<long>(gridDim.x) * 256;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8; i += 2 * stride) {
    const long j = i + stride;
    const F8 a = load8(x + i * 8);
    F8 b = a;
    if (j < n8) b = load8(x + j * 8);
    F8 oa, ob;
#pragma unroll
    for (int e = 0; e < 8; ++e) { oa.v[e] = gelu_f(a.v[e]); ob.v[e] = gelu_f(b.v[e]); }
    store8(y + i * 8, oa);
    if (j < n8) store8(y + j * 8, ob);
  }
}

inline int ln_wide_group(int C) {              // C = 24 * G with G a power of two in [2, 32]
  if (C % 24 != 0) return 0;
  const int g = C / 24;
  return (g >= 2 && g <= 32 && (g & (g - 1)) == 0) ? g : 0;
}

template <typename TX, typename TY>
int launch_ln_fwd(const TX* x, const float* w, const float* b, float eps, TY* y, float* mean, float* rstd, long M, int C,
                  int gelu, hipStream_t s, int pH = 0, int pW = 0) {
#define LN_FWD(G, NV)                                                                                          \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > 16384) nb = 16384;                                                 \
    hipLaunchKernelGGL((layernorm_fwd_kernel<TX, TY, G, NV>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, x, w, \
                       b, eps, y, mean, rstd, M, C, gelu);                                                     \
    return launch_status();                                                                                    \
  }
  constexpr int wide_on = 1;     // 0: 4-channel kernels everywhere
  if (const int g = wide_on ? ln_wide_group(C) : 0) {
#define LN_FWD_W(G)                                                                                            \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > 16384) nb = 16384;                                                 \
    hipLaunchKernelGGL((layernorm_fwd_wide_kernel<TX, TY, G>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, x, w, \
                       b, eps, y, mean, rstd, M, C, gelu, pH, pW);                                             \
    return launch_status();                                                                                    \
  }
    if (g == 2) LN_FWD_W(2)
    if (g == 4) LN_FWD_W(4)
    if (g == 8) LN_FWD_W(8)
    if (g == 16) LN_FWD_W(16)
    if (g == 32) LN_FWD_W(32)
#undef LN_FWD_W
  }
  if (pW) return APGD_ERR_ARG;                              // the patch form exists for the wide kernels only
  if (C <= 64) LN_FWD(16, 1)
  if (C <= 128) LN_FWD(32, 1)
  if (C <= 256) LN_FWD(64, 1)
  if (C <= 512) LN_FWD(64, 2)
  if (C <= 1024) LN_FWD(64, 4)
  if (C <= 2048) LN_FWD(64, 8)
#undef LN_FWD
  return APGD_ERR_SIZE;
}

template <typename TD, typename TX, typename TO>
int launch_ln_bwd(const TD* dy, const TX* x, const float* w, const float* b, const float* mean, const float* rstd, TO* dx,
                  float* ws, long M, int C, int gelu, int* nblocks, hipStream_t s, int pH = 0, int pW = 0,
                  const float* add = nullptr) {
#define LN_BWD(G, NV)                                                                                          \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > kLnBwdBlocks) nb = kLnBwdBlocks;                                   \
    *nblocks = static_cast<int>(nb);                                                                           \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TD, TX, TO, G, NV>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, dy, \
                       x, w, b, mean, rstd, dx, ws, M, C, gelu, add);                                          \
    return launch_status();                                                                                    \
  }
  constexpr int wide_on = 1;
  if (const int g = wide_on ? ln_wide_group(C) : 0) {
#define LN_BWD_W(G)                                                                                            \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > kLnBwdBlocks) nb = kLnBwdBlocks;                                   \
    *nblocks = static_cast<int>(nb);                                                                           \
    if (ws)                                                                                                    \
      hipLaunchKernelGGL((layernorm_bwd_wide_kernel<TD, TX, TO, G, true>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, dy, \
                         x, w, b, mean, rstd, dx, ws, M, C, gelu, pH, pW, add);                                \
    else                                                                                                       \
      hipLaunchKernelGGL((layernorm_bwd_wide_kernel<TD, TX, TO, G, false>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, dy, \
                         x, w, b, mean, rstd, dx, ws, M, C, gelu, pH, pW, add);                                \
    return launch_status();                                                                                    \
  }
    if (g == 2) LN_BWD_W(2)
    if (g == 4) LN_BWD_W(4)
    if (g == 8) LN_BWD_W(8)
    if (g == 16) LN_BWD_W(16)
    if (g == 32) LN_BWD_W(32)
#undef LN_BWD_W
  }
  if (pW) return APGD_ERR_ARG;
  if (C <= 64) LN_BWD(16, 1)
  if (C <= 128) LN_BWD(32, 1)
  if (C <= 256) LN_BWD(64, 1)
  if (C <= 512) LN_BWD(64, 2)
  if (C <= 1024) LN_BWD(64, 4)
  if (C <= 2048) LN_BWD(64, 8)
#undef LN_BWD
  return APGD_ERR_SIZE;
}

}  // namespace

// =================================================================================== C ABI
extern "C" {

int cnx_dwconv7x7_nhwc(const void* x, int x_dtype, const float* w49c, const float* bias, const float* add, void* out,
                       int out_dtype, int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0) return APGD_ERR_SIZE;
  if (N == 0) return APGD_OK;
  if (!x || !w49c || !out) return APGD_ERR_NULL;
  if (C % 4 != 0) return APGD_ERR_ARG;
  if ((x_dtype != APGD_F32 && x_dtype != APGD_BF16) || (out_dtype != APGD_F32 && out_dtype != APGD_BF16))
    return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  const bool all_f32 = x_dtype == APGD_F32 && out_dtype == APGD_F32;
  {
    // bf16 operands: the register-window kernel (dwwin_kernels.hip, round 4) takes every shape with C % 32 == 0
    const int rc = dw_win_launch(x, x_dtype, w49c, bias, add, out, out_dtype, N, H, W, C, flip, s);
    if (rc != -1) return rc;
  }
#if DW_ABLATE
  static const int dw_dbg = getenv("APGD_DW_DBG") ? atoi(getenv("APGD_DW_DBG")) : 0;   // timing experiments: ablation builds only
#else
  constexpr int dw_dbg = 0;
#endif
  // The packed-dot kernels fuse "+ add" only into an fp32 result.  A bf16 result with an add operand (the residual gradient
  // of a block whose input is bf16) goes to the strip kernel below, which honours it for every output type - never silently
  // dropped.
  const bool add_into_bf16 = add != nullptr && out_dtype == APGD_BF16;
  if (add_into_bf16 && x_dtype == APGD_BF16) {
    // input-gradient call of a block whose input is bf16 (first block of a stage): fp32 staging of the result tile, the sum with
    // the residual gradient rounded once
    DwRoll rp;
    DwMulti mp;
    if (dw_roll_plan(H, W, C, 2, 4, &rp)) {
      const dim3 grid(static_cast<unsigned>(static_cast<long>(N) * (C / kDC) * rp.n_seg)), block(rp.threads);
#define DWRA_LAUNCH(UU, MT)                                                                                           \
  {                                                                                                                   \
    auto kfn = dwconv7x7_roll_kernel<uint16_t, uint16_t, UU, true, MT>;                                               \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, rp.lds, s, static_cast<const uint16_t*>(x), w49c, bias, add,                 \
                       static_cast<uint16_t*>(out), H, W, C, flip, rp.rs, rp.n_seg);                                   \
  }
      if (rp.threads > 256) DWRA_LAUNCH(1, kRollMaxThreads)
      else if (rp.units == 1) DWRA_LAUNCH(1, 256) else DWRA_LAUNCH(2, 256)
#undef DWRA_LAUNCH
      return launch_status();
    }
    if (dw_multi_plan(N, H, W, C, 2, 4, &mp)) {
      const dim3 grid(static_cast<unsigned>(static_cast<long>((N + mp.ipw - 1) / mp.ipw) * (C / kDC))), block(mp.threads);
#define DWMA_LAUNCH(UU)                                                                                               \
  {                                                                                                                   \
    auto kfn = dwconv7x7_multi_kernel<uint16_t, uint16_t, UU, true>;                                                  \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, mp.lds, s, static_cast<const uint16_t*>(x), w49c, bias, add,                 \
                       static_cast<uint16_t*>(out), N, H, W, C, flip, mp.n_sr, mp.ipw);                                \
  }
      if (mp.ut == 5) DWMA_LAUNCH(5) else DWMA_LAUNCH(7)
#undef DWMA_LAUNCH
      return launch_status();
    }
  }
  if (!all_f32 && !add_into_bf16) {
    DwRoll rp;
    if (dw_roll_plan(H, W, C, x_dtype == APGD_F32 ? 4 : 2, out_dtype == APGD_F32 ? 4 : 2, &rp)) {
      const dim3 grid(static_cast<unsigned>(static_cast<long>(N) * (C / kDC) * rp.n_seg)), block(rp.threads);
#define DWR_LAUNCH(TI, TO, UU, MT)                                                                                    \
  {                                                                                                                   \
    auto kfn = dwconv7x7_roll_kernel<TI, TO, UU, false, MT>;                                                          \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, rp.lds, s, static_cast<const TI*>(x), w49c, bias, add, static_cast<TO*>(out), H, W, \
                       C, flip, rp.rs, rp.n_seg);                                                                      \
  }
#define DWR_DISPATCH(TI, TO)                                                                                          \
  {                                                                                                                   \
    if (rp.threads > 256) { if constexpr (sizeof(TI) == 2) DWR_LAUNCH(TI, TO, 1, kRollMaxThreads) else return APGD_ERR_ARG; } /* plan: bf16 inputs only */ \
    else if (rp.units == 1) DWR_LAUNCH(TI, TO, 1, 256) else DWR_LAUNCH(TI, TO, 2, 256)                                \
  }
      if (x_dtype == APGD_F32) DWR_DISPATCH(float, uint16_t)
      else if (out_dtype == APGD_F32) DWR_DISPATCH(uint16_t, float)
      else DWR_DISPATCH(uint16_t, uint16_t)
#undef DWR_DISPATCH
#undef DWR_LAUNCH
      return launch_status();
    }
    DwMulti mp;
    // (the input gradient with the fp32 residual gradient added, bf16 -> fp32 + add, at 14x14: the multi-image kernel holds the add
    //  operand of the next image in registers on top of its filter and prefetch registers - one wavefront per SIMD - and the
    //  whole-image tile kernel below, four per SIMD, is ahead since its staging was spread over the workgroup: 68 vs 81 us at
    //  14x14x384 x 256, 42 vs 55 us x 128; at 7x7 the two are level)
    const bool tile_ahead = out_dtype == APGD_F32 && add != nullptr && H * W >= 144;
    if (!tile_ahead && dw_multi_plan(N, H, W, C, x_dtype == APGD_F32 ? 4 : 2, out_dtype == APGD_F32 ? 4 : 2, &mp)) {
      const dim3 grid(static_cast<unsigned>(static_cast<long>((N + mp.ipw - 1) / mp.ipw) * (C / kDC))), block(mp.threads);
#define DWM_LAUNCH(TI, TO, UU)                                                                                        \
  {                                                                                                                   \
    auto kfn = dwconv7x7_multi_kernel<TI, TO, UU>;                                                                    \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, mp.lds, s, static_cast<const TI*>(x), w49c, bias, add, static_cast<TO*>(out), N, H, \
                       W, C, flip, mp.n_sr, mp.ipw);                                                                   \
  }
#define DWM_DISPATCH(TI, TO) { if (mp.ut == 5) DWM_LAUNCH(TI, TO, 5) else DWM_LAUNCH(TI, TO, 7) }
      if (out_dtype == APGD_F32) DWM_DISPATCH(uint16_t, float)       // the plan only accepts bf16 inputs
      else DWM_DISPATCH(uint16_t, uint16_t)
#undef DWM_DISPATCH
#undef DWM_LAUNCH
      return launch_status();
    }
    DwDot dp;
    if (dw_dot2_plan(H, W, C, &dp)) {
      const int tiles_h = (H + dp.th - 1) / dp.th;
      constexpr int ipw_env = 0;     // tuning experiments only
      const int ipw = tiles_h == 1 ? (ipw_env > 0 ? ipw_env : (H * W <= 64 ? 2 : 1)) : 1;     // measured: 7x7 40 -> 36 us, 14x14 no gain
      const dim3 grid(static_cast<unsigned>(((N + ipw - 1) / ipw) * tiles_h * (C / kDC))), block(dp.threads);
#define DWD_LAUNCH(TI, TO)                                                                                            \
  {                                                                                                                   \
    auto kfn = dwconv7x7_dot2_kernel<TI, TO>;                                                                         \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, dp.lds, s, static_cast<const TI*>(x), w49c, bias, add, static_cast<TO*>(out),    \
                       static_cast<long>(N), H, W, C, flip, dp.th, tiles_h, ipw, dw_dbg);                              \
  }
      if (x_dtype == APGD_F32) DWD_LAUNCH(float, uint16_t)
      else if (out_dtype == APGD_F32) DWD_LAUNCH(uint16_t, float)
      else DWD_LAUNCH(uint16_t, uint16_t)
#undef DWD_LAUNCH
      return launch_status();
    }
  }
  {
    DwTile tp;
    if (all_f32 && dw_tile_plan(H, W, C, 4, &tp)) {
      const int tiles_h = (H + tp.th - 1) / tp.th;
      const dim3 grid(static_cast<unsigned>(N * tiles_h), C / tp.cc), block(tp.threads);
#define DWT_LAUNCH(TI, TO, TS, CCV)                                                                                   \
  {                                                                                                                   \
    auto kfn = dwconv7x7_tile_kernel<TI, TO, TS, CCV>;                                                                \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, tp.lds, s, static_cast<const TI*>(x), w49c, bias, add, static_cast<TO*>(out), H, W, \
                       C, flip, tp.th, tiles_h);                                                                       \
  }
#define DWT_TYPES(CCV) DWT_LAUNCH(float, float, float, CCV)
      if (tp.cc == 32) { DWT_TYPES(32) } else if (tp.cc == 64) { DWT_TYPES(64) } else { DWT_TYPES(128) }
#undef DWT_TYPES
#undef DWT_LAUNCH
      return launch_status();
    }
  }
  const int cc = C < kCC ? C : kCC;
  const int lc = cc / 4;
  const int py = 240 / lc > 0 ? 240 / lc : 1;
  const int tiles_h = (H + kRH - 1) / kRH, tiles_w = (W + kTW - 1) / kTW;
  const long n_strips = static_cast<long>(N) * tiles_h * tiles_w;
  const dim3 block(lc, py), grid(static_cast<unsigned>((n_strips + py - 1) / py), (C + kCC - 1) / kCC);
  const size_t lds = static_cast<size_t>(49) * cc * sizeof(float);
#define DW_LAUNCH(TI, TO)                                                                                       \
  hipLaunchKernelGGL((dwconv7x7_kernel<TI, TO>), grid, block, lds, s, static_cast<const TI*>(x), w49c, bias, add, \
                     static_cast<TO*>(out), H, W, C, flip, tiles_h, tiles_w, n_strips)
  if (x_dtype == APGD_F32 && out_dtype == APGD_F32) DW_LAUNCH(float, float);
  else if (x_dtype == APGD_F32) DW_LAUNCH(float, uint16_t);
  else if (out_dtype == APGD_F32) DW_LAUNCH(uint16_t, float);
  else DW_LAUNCH(uint16_t, uint16_t);
#undef DW_LAUNCH
  return launch_status();
}

int64_t cnx_dwconv7x7_wgrad_ws_floats(int32_t C) { return static_cast<int64_t>(kWgradBlocks) * 50 * C; }

int cnx_dwconv7x7_wgrad_nhwc(const void* x, int x_dtype, const void* dy, int dy_dtype, float* dw49c, float* dbias,
                             float* ws, int64_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0) return APGD_ERR_SIZE;
  if (!x || !dy || !dw49c || !ws) return APGD_ERR_NULL;
  if (C % 4 != 0) return APGD_ERR_ARG;
  if ((x_dtype != APGD_F32 && x_dtype != APGD_BF16) || (dy_dtype != APGD_F32 && dy_dtype != APGD_BF16))
    return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  if (N == 0) return APGD_OK;
  if (dy_dtype == APGD_BF16 && C % kDC == 0 && W <= 128) {
    // autocast path: bf16 output gradient; an fp32 x is rounded to bf16 (as the convolution itself did); packed dot products
    const int D2 = (W + 1) / 2, P2 = D2 + 3;
    {
      // register-window form (dwwin_kernels.hip, round 4): every map at least 7 columns wide
      const int parts = dw_win_wgrad_launch(x, x_dtype, dy, ws, kWgradBlocks, N, H, W, C, s);
      if (parts < 0) return -(parts + 1);
      if (parts > 0) {
        const int len = 50 * C;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 15) / 16), dim3(256), 0, s, ws, dw49c, dbias, 49 * C, len, parts);
        return launch_status();
      }
    }
    constexpr int wroll = 1;        // 0: tile kernel
    if (wroll && P2 * (kDC / 4) <= 256) {
      constexpr int rs_env = 0;         // tuning experiments only
      constexpr int im_env = 0;
      int rs = rs_env > 0 ? rs_env : (H >= 56 ? 28 : H);
      rs = ((rs + 3) / 4) * 4;
      const int n_seg = (H + rs - 1) / rs;
      int imgs = im_env > 0 ? im_env : (H * W >= 784 ? 1 : (H * W >= 196 ? 2 : 4));
      while (((N + imgs - 1) / imgs) * n_seg > kWgradBlocks) ++imgs;
      const int nparts = static_cast<int>((N + imgs - 1) / imgs) * n_seg;
      const size_t lds = (static_cast<size_t>(10) * P2 + static_cast<size_t>(4) * D2 + 1) * kDC * 4;   // + 1 pair: the look-ahead read
      const dim3 grid(static_cast<unsigned>(nparts) * (C / kDC)), block(256);
#define WGR_LAUNCH(TX)                                                                                                \
  {                                                                                                                   \
    auto kfn = dwconv7x7_wgrad_roll_kernel<TX>;                                                                       \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, static_cast<const TX*>(x), static_cast<const uint16_t*>(dy), ws,     \
                       static_cast<long>(N), H, W, C, rs, n_seg, imgs);                                                \
  }
      if (x_dtype == APGD_F32) WGR_LAUNCH(float)
      else WGR_LAUNCH(uint16_t)
#undef WGR_LAUNCH
      const int len = 50 * C;
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 15) / 16), dim3(256), 0, s, ws, dw49c, dbias, 49 * C, len, nparts);
      return launch_status();
    }
    int th = H < 8 ? H : 8;
    auto lds_of = [&](int rows) { return (static_cast<size_t>(rows + 6) * P2 + static_cast<size_t>(rows) * D2) * kDC * 4; };
    while (th > 1 && lds_of(th) > 56 * 1024) --th;
    const int tiles_h = (H + th - 1) / th;
    int imgs = H * W >= 784 ? 1 : (H * W >= 196 ? 2 : 4);
    while ((N + imgs - 1) / imgs > kWgradBlocks) ++imgs;
    const int nparts = static_cast<int>((N + imgs - 1) / imgs);
    const dim3 grid(nparts, C / kDC), block(256);
#define WGD_LAUNCH(TX, TD)                                                                                            \
  {                                                                                                                   \
    auto kfn = dwconv7x7_wgrad_dot2_kernel<TX, TD>;                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, lds_of(th), s, static_cast<const TX*>(x), static_cast<const TD*>(dy), ws,     \
                       static_cast<long>(N), H, W, C, th, tiles_h, imgs);                                              \
  }
    if (x_dtype == APGD_F32) WGD_LAUNCH(float, uint16_t)
    else WGD_LAUNCH(uint16_t, uint16_t)
#undef WGD_LAUNCH
    const int len = 50 * C;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 15) / 16), dim3(256), 0, s, ws, dw49c, dbias, 49 * C, len, nparts);
    return launch_status();
  }
  const int cc = C < kCC ? C : kCC;
  const int tiles_w = (W + kTW - 1) / kTW;
  const long n_strips = static_cast<long>(N) * H * tiles_w;
  int nb = kWgradBlocks;
  if (n_strips < nb) nb = static_cast<int>(n_strips > 0 ? n_strips : 1);
  const dim3 block(cc / 4, 7), grid(nb, (C + kCC - 1) / kCC);
#define WG_LAUNCH(TX, TD)                                                                                   \
  hipLaunchKernelGGL((dwconv7x7_wgrad_kernel<TX, TD>), grid, block, 0, s, static_cast<const TX*>(x),          \
                     static_cast<const TD*>(dy), ws, H, W, C, tiles_w, n_strips)
  if (x_dtype == APGD_F32 && dy_dtype == APGD_F32) WG_LAUNCH(float, float);
  else if (x_dtype == APGD_F32) WG_LAUNCH(float, uint16_t);
  else if (dy_dtype == APGD_F32) WG_LAUNCH(uint16_t, float);
  else WG_LAUNCH(uint16_t, uint16_t);
#undef WG_LAUNCH
  const int len = 50 * C;
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 15) / 16), dim3(256), 0, s, ws, dw49c, dbias, 49 * C, len, nb);
  return launch_status();
}

static int layernorm_fwd_impl(const void* x, int x_dtype, const float* weight, const float* bias, float eps, void* y,
                              int y_dtype, float* mean, float* rstd, int64_t M, int32_t C, int32_t gelu, int pH, int pW,
                              void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!x || !weight || !bias || !y) return APGD_ERR_NULL;
  if ((mean == nullptr) != (rstd == nullptr)) return APGD_ERR_ARG;
  if (C % 4 != 0) return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (x_dtype == APGD_F32 && y_dtype == APGD_F32)
    return launch_ln_fwd(static_cast<const float*>(x), weight, bias, eps, static_cast<float*>(y), mean, rstd, M, C, gelu, s, pH, pW);
  if (x_dtype == APGD_F32 && y_dtype == APGD_BF16)
    return launch_ln_fwd(static_cast<const float*>(x), weight, bias, eps, static_cast<uint16_t*>(y), mean, rstd, M, C, gelu, s, pH, pW);
  if (x_dtype == APGD_BF16 && y_dtype == APGD_F32)
    return launch_ln_fwd(static_cast<const uint16_t*>(x), weight, bias, eps, static_cast<float*>(y), mean, rstd, M, C, gelu, s, pH, pW);
  if (x_dtype == APGD_BF16 && y_dtype == APGD_BF16)
    return launch_ln_fwd(static_cast<const uint16_t*>(x), weight, bias, eps, static_cast<uint16_t*>(y), mean, rstd, M, C, gelu, s, pH, pW);
  return APGD_ERR_DTYPE;
}

int cnx_layernorm_fwd(const void* x, int x_dtype, const float* weight, const float* bias, float eps, void* y,
                      int y_dtype, float* mean, float* rstd, int64_t M, int32_t C, int32_t gelu, void* stream) {
  return layernorm_fwd_impl(x, x_dtype, weight, bias, eps, y, y_dtype, mean, rstd, M, C, gelu, 0, 0, stream);
}

int cnx_layernorm_fwd_patch2(const void* x, int x_dtype, const float* weight, const float* bias, float eps, void* y,
                             int y_dtype, float* mean, float* rstd, int64_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if ((H | W) & 1) return APGD_ERR_ARG;
  if (!ln_wide_group(C)) return APGD_ERR_ARG;
  return layernorm_fwd_impl(x, x_dtype, weight, bias, eps, y, y_dtype, mean, rstd, N * H * W, C, 0, H, W, stream);
}

int64_t cnx_layernorm_bwd_ws_floats(int32_t C) { return static_cast<int64_t>(kLnBwdBlocks) * 2 * C; }

static int layernorm_bwd_impl(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* bias,
                              const float* mean, const float* rstd, void* dx, int dx_dtype, float* dweight, float* dbias,
                              float* ws, int64_t M, int32_t C, int32_t gelu, int pH, int pW, void* stream,
                              const float* add = nullptr) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!dy || !x || !weight || !mean || !rstd || !dx) return APGD_ERR_NULL;
  if (add && dx_dtype != APGD_F32) return APGD_ERR_DTYPE;          // the skip gradient is summed in fp32 into an fp32 result
  if (gelu && !bias) return APGD_ERR_NULL;
  if (dweight && (!dbias || !ws)) return APGD_ERR_NULL;
  if (C % 4 != 0) return APGD_ERR_ARG;
  if ((dy_dtype | x_dtype | dx_dtype) & ~1) return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  float* wsp = dweight ? ws : nullptr;
  int nb = 0, rc = APGD_ERR_DTYPE;
#define LNB(TD, TX, TO)                                                                                          \
  rc = launch_ln_bwd(static_cast<const TD*>(dy), static_cast<const TX*>(x), weight, bias, mean, rstd,            \
                     static_cast<TO*>(dx), wsp, M, C, gelu, &nb, s, pH, pW, add)
  const int key = dy_dtype * 4 + x_dtype * 2 + dx_dtype;
  switch (key) {
    case 0: LNB(float, float, float); break;
    case 1: LNB(float, float, uint16_t); break;
    case 2: LNB(float, uint16_t, float); break;
    case 3: LNB(float, uint16_t, uint16_t); break;
    case 4: LNB(uint16_t, float, float); break;
    case 5: LNB(uint16_t, float, uint16_t); break;
    case 6: LNB(uint16_t, uint16_t, float); break;
    case 7: LNB(uint16_t, uint16_t, uint16_t); break;
    default: return APGD_ERR_DTYPE;
  }
#undef LNB
  if (rc != APGD_OK) return rc;
  if (dweight) {
    const int len = 2 * C;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 15) / 16), dim3(256), 0, s, ws, dweight, dbias, C, len, nb);
    return launch_status();
  }
  return APGD_OK;
}

int cnx_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* bias,
                      const float* mean, const float* rstd, void* dx, int dx_dtype, float* dweight, float* dbias,
                      float* ws, int64_t M, int32_t C, int32_t gelu, void* stream) {
  return layernorm_bwd_impl(dy, dy_dtype, x, x_dtype, weight, bias, mean, rstd, dx, dx_dtype, dweight, dbias, ws, M, C, gelu, 0, 0,
                            stream);
}

int cnx_layernorm_bwd_add(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* bias,
                          const float* mean, const float* rstd, const float* add, void* dx, int dx_dtype, float* dweight,
                          float* dbias, float* ws, int64_t M, int32_t C, int32_t gelu, void* stream) {
  return layernorm_bwd_impl(dy, dy_dtype, x, x_dtype, weight, bias, mean, rstd, dx, dx_dtype, dweight, dbias, ws, M, C, gelu, 0, 0,
                            stream, add);
}

int cnx_layernorm_bwd_patch2(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* mean,
                             const float* rstd, void* dx, int dx_dtype, float* dweight, float* dbias, float* ws, int64_t N,
                             int32_t H, int32_t W, int32_t C, void* stream) {
  if (N < 0 || H <= 0 || W <= 0) return APGD_ERR_SIZE;
  if ((H | W) & 1) return APGD_ERR_ARG;
  if (!ln_wide_group(C)) return APGD_ERR_ARG;
  return layernorm_bwd_impl(dy, dy_dtype, x, x_dtype, weight, nullptr, mean, rstd, dx, dx_dtype, dweight, dbias, ws, N * H * W, C, 0,
                            H, W, stream);
}

int cnx_sum_parts_bf16(const void* parts, float* out, int64_t S, int64_t L, void* stream) {
  if (S < 0 || L < 0 || L % 8 != 0) return APGD_ERR_SIZE;
  if (L == 0) return APGD_OK;
  if (!parts || !out) return APGD_ERR_NULL;
  hipLaunchKernelGGL(sum_parts_bf16_kernel, dim3(static_cast<unsigned>((L / 8 + 63) / 64)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint16_t*>(parts), out, static_cast<long>(S), static_cast<long>(L));
  return launch_status();
}

int64_t cnx_colsum_ws_floats(int32_t n_cols) { return n_cols > 0 ? static_cast<int64_t>(col_parts(n_cols)) * 2 * n_cols : 0; }

int cnx_scale_residual(const void* x, int x_dtype, const void* y, const float* gamma, void* out, int out_dtype, int64_t M,
                       int32_t C, void* stream) {
  if (M < 0 || C <= 0 || C % 4 != 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!x || !y || !out) return APGD_ERR_NULL;
  if ((x_dtype != APGD_F32 && x_dtype != APGD_BF16) || (out_dtype != APGD_F32 && out_dtype != APGD_BF16)) return APGD_ERR_DTYPE;
  int cq;
  const dim3 grid = col_grid(C, M, 16384, &cq), block(256);
  hipStream_t s = as_stream(stream);
  const auto* yy = static_cast<const uint16_t*>(y);
#define SR(TX, TO) hipLaunchKernelGGL((scale_residual_kernel<TX, TO>), grid, block, 0, s, static_cast<const TX*>(x), yy, gamma, static_cast<TO*>(out), static_cast<long>(M), C, cq)
  if (x_dtype == APGD_F32 && out_dtype == APGD_F32) SR(float, float);
  else if (x_dtype == APGD_F32) SR(float, uint16_t);
  else if (out_dtype == APGD_F32) SR(uint16_t, float);
  else SR(uint16_t, uint16_t);
#undef SR
  return launch_status();
}

int cnx_scale_residual_bwd(const void* g, int g_dtype, const void* y, const float* gamma, void* dos, float* dgamma, float* db2,
                           float* ws, int64_t M, int32_t C, void* stream) {
  if (M < 0 || C <= 0 || C % 4 != 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!g || (!dos && !dgamma)) return APGD_ERR_NULL;
  if ((dgamma == nullptr) != (db2 == nullptr)) return APGD_ERR_NULL;          // both sums or neither
  if (dgamma && !ws) return APGD_ERR_NULL;
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  const bool sums = dgamma != nullptr;
  int cq;
  const dim3 grid = col_grid(C, M, sums ? col_parts(C) : 16384, &cq), block(256);
  const int parts = col_lanes(cq, grid);
  hipStream_t s = as_stream(stream);
  const auto* yy = static_cast<const uint16_t*>(y);          // NULL: dgamma comes out as zeros
  if (g_dtype == APGD_F32)
    hipLaunchKernelGGL(scale_residual_bwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(g), yy, gamma,
                       static_cast<uint16_t*>(dos), sums ? ws : nullptr, static_cast<long>(M), C, cq);
  else
    hipLaunchKernelGGL(scale_residual_bwd_kernel<uint16_t>, grid, block, 0, s, static_cast<const uint16_t*>(g), yy, gamma,
                       static_cast<uint16_t*>(dos), sums ? ws : nullptr, static_cast<long>(M), C, cq);
  if (sums)
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, s, ws, dgamma, db2, C, 2 * C,
                       static_cast<int>(M < parts ? M : parts));
  return launch_status();
}

int cnx_block_dgamma(const float* w2, const float* dw2, const float* b2, const float* db2, const float* gamma, const void* g,
                     int g_dtype, const void* y2, const void* h_tiles, float* dgamma, int64_t M, int32_t C, int32_t Hd, void* stream) {
  if (M < 0 || C <= 0 || Hd <= 0) return APGD_ERR_SIZE;
  if (!w2 || !dw2 || !gamma || !g || (!y2 && !h_tiles) || !dgamma || (b2 && !db2)) return APGD_ERR_NULL;
  if (!y2 && (M % 32 != 0 || Hd % 32 != 0)) return APGD_ERR_ARG;                // the tile form holds whole 32 x 32 tiles
  if (g_dtype != APGD_F32 && g_dtype != APGD_BF16) return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  if (g_dtype == APGD_F32)
    hipLaunchKernelGGL(block_dgamma_kernel<float>, dim3(C), dim3(256), 0, s, w2, dw2, b2, db2, gamma, static_cast<const float*>(g),
                       static_cast<const uint16_t*>(y2), static_cast<const uint16_t*>(h_tiles), dgamma, static_cast<long>(M), C, Hd);
  else
    hipLaunchKernelGGL(block_dgamma_kernel<uint16_t>, dim3(C), dim3(256), 0, s, w2, dw2, b2, db2, gamma, static_cast<const uint16_t*>(g),
                       static_cast<const uint16_t*>(y2), static_cast<const uint16_t*>(h_tiles), dgamma, static_cast<long>(M), C, Hd);
  return launch_status();
}

int64_t cnx_block_dln_ws_floats(int32_t C) { return C > 0 ? static_cast<int64_t>(kDlnParts) * C : 0; }

int cnx_block_dln(const float* w1, const float* dw1, const float* db1, const float* ln_w, const float* ln_b, const void* da,
                  const void* dhpre_tiles, const void* u, const float* mean, const float* rstd, float* dlw, float* dlb, float* ws, int64_t M,
                  int32_t C, int32_t Hd, void* stream) {
  if (M < 0 || C <= 0 || Hd <= 0) return APGD_ERR_SIZE;
  if (!w1 || !dw1 || !db1 || !ln_w || !ln_b || (!da && !dhpre_tiles) || !u || !mean || !rstd || !dlw || !dlb) return APGD_ERR_NULL;
  if (!da && (M % 32 != 0 || Hd % 32 != 0)) return APGD_ERR_ARG;
  const size_t lds = da ? 0 : static_cast<size_t>(Hd) * kDlnChunk * sizeof(float);
  if (C > 2048 || lds > 96 * 1024) ws = nullptr;                                // beyond the direct kernel's channel list / LDS image: serial sums
  if (ws) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_dln_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_done = true;
    }
    hipLaunchKernelGGL(block_dln_direct_kernel, dim3(kDlnParts), dim3(256), lds, as_stream(stream), w1, ln_w, ln_b,
                       static_cast<const uint16_t*>(da), static_cast<const uint16_t*>(dhpre_tiles), static_cast<const uint16_t*>(u), mean, rstd,
                       ws, static_cast<long>(M), C, Hd);
  }
  hipLaunchKernelGGL(block_dln_kernel, dim3((C + 31) / 32), dim3(1024), 0, as_stream(stream), w1, dw1, db1, ln_w, ln_b, static_cast<const uint16_t*>(da),
                     static_cast<const uint16_t*>(dhpre_tiles), static_cast<const uint16_t*>(u), mean, rstd, dlw, dlb, ws, kDlnParts, static_cast<long>(M), C, Hd);
  return launch_status();
}

int cnx_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
  if (n < 0 || n % 8 != 0) return APGD_ERR_SIZE;
  if (n == 0) return APGD_OK;
  if (!x || !y) return APGD_ERR_NULL;
  const long n8 = n / 8;
  long nb = (n8 + 511) / 512; if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint16_t*>(x), static_cast<uint16_t*>(y), n8);
  return launch_status();
}

int cnx_gelu_bwd_colsum(const void* dh, const void* hpre, void* dhpre, float* db1, float* ws, int64_t M, int32_t N, void* stream) {
  if (M < 0 || N <= 0 || N % 4 != 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!dh || !hpre || !dhpre) return APGD_ERR_NULL;
  if (db1 && !ws) return APGD_ERR_NULL;
  int cq;
  const dim3 grid = col_grid(N, M, db1 ? col_parts(N) : 16384, &cq), block(256);
  const int parts = col_lanes(cq, grid);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, grid, block, 0, s, static_cast<const uint16_t*>(dh), static_cast<const uint16_t*>(hpre),
                     static_cast<uint16_t*>(dhpre), db1 ? ws : nullptr, static_cast<long>(M), N, cq);
  if (db1)
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((N + 15) / 16), dim3(256), 0, s, ws, db1, static_cast<float*>(nullptr), N, N,
                       static_cast<int>(M < parts ? M : parts));
  return launch_status();
}

}  // extern "C"
