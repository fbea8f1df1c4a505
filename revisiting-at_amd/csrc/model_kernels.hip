// model_kernels.hip — gfx950 kernels for the ConvNeXt block / ConvStem pieces of the hot path
// (C ABI: include/convnext_hip.h).  Reference arithmetic: /root/reference/models/convnext.py:28-41
// (depthwise 7x7 + LayerNorm) and /root/reference/utils_architecture.py:76-81 (+GELU in the stems).
//
// All of these are bandwidth-bound (no contraction over channels): channels-last rows, 16-byte
// per-lane accesses coalesced along C, fp32 accumulation, wavefront-64 shuffles + LDS for the
// per-row statistics, deterministic two-stage reductions for parameter gradients.  No MFMA here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

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
  extern __shared__ float wl[];                       // [49][cc]
  const int cbase = blockIdx.y * kCC;
  const int cc = min(kCC, C - cbase);
  const int lc = threadIdx.x, py = threadIdx.y;
  const int nthr = blockDim.x * blockDim.y, tid = py * blockDim.x + lc;
  for (int i = tid; i < 49 * cc; i += nthr) {
    const int tap = i / cc, c = i - tap * cc;
    wl[(flip ? 48 - tap : tap) * cc + c] = w49c[tap * C + cbase + c];
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

// out[j] = sum_p ws[p*len + j]: 64 outputs per block, the parts split over 4 thread groups whose
// partial sums are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ ws, float* __restrict__ out0,
                                                           float* __restrict__ out1, int split, int len, int nparts) {
  __shared__ float part[4][64];
  const int jl = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  float s = 0.f;
  if (j < len)
    for (int p = pg; p < nparts; p += 4) s += ws[static_cast<long>(p) * len + j];
  part[pg][jl] = s;
  __syncthreads();
  if (pg == 0 && j < len) {
    s = (part[0][jl] + part[1][jl]) + (part[2][jl] + part[3][jl]);
    if (j < split) out0[j] = s; else if (out1) out1[j - split] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over C of [M, C] rows (+ optional exact GELU).  GROUP lanes cooperate on one row,
// 64/GROUP rows per wavefront, 4 channels per lane-chunk, up to NV chunks per lane in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * expf(-0.5f * z * z);
  return cdf + z * pdf;
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

constexpr int kLnBwdBlocks = 1024;

template <typename TD, typename TX, typename TO, int GROUP, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TO* __restrict__ dx,
                                                            float* __restrict__ ws, long M, int C, int gelu) {
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

template <typename TX, typename TY>
int launch_ln_fwd(const TX* x, const float* w, const float* b, float eps, TY* y, float* mean, float* rstd, long M, int C,
                  int gelu, hipStream_t s) {
#define LN_FWD(G, NV)                                                                                          \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > 16384) nb = 16384;                                                 \
    hipLaunchKernelGGL((layernorm_fwd_kernel<TX, TY, G, NV>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, x, w, \
                       b, eps, y, mean, rstd, M, C, gelu);                                                     \
    return launch_status();                                                                                    \
  }
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
                  float* ws, long M, int C, int gelu, int* nblocks, hipStream_t s) {
#define LN_BWD(G, NV)                                                                                          \
  {                                                                                                            \
    const long rpb = 256 / G;                                                                                  \
    long nb = (M + rpb - 1) / rpb; if (nb > kLnBwdBlocks) nb = kLnBwdBlocks;                                   \
    *nblocks = static_cast<int>(nb);                                                                           \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TD, TX, TO, G, NV>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, s, dy, \
                       x, w, b, mean, rstd, dx, ws, M, C, gelu);                                               \
    return launch_status();                                                                                    \
  }
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
  const int cc = C < kCC ? C : kCC;
  const int lc = cc / 4;
  const int py = 240 / lc > 0 ? 240 / lc : 1;
  const int tiles_h = (H + kRH - 1) / kRH, tiles_w = (W + kTW - 1) / kTW;
  const long n_strips = static_cast<long>(N) * tiles_h * tiles_w;
  const dim3 block(lc, py), grid(static_cast<unsigned>((n_strips + py - 1) / py), (C + kCC - 1) / kCC);
  const size_t lds = static_cast<size_t>(49) * cc * sizeof(float);
  hipStream_t s = as_stream(stream);
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
  const int cc = C < kCC ? C : kCC;
  const int tiles_w = (W + kTW - 1) / kTW;
  const long n_strips = static_cast<long>(N) * H * tiles_w;
  int nb = kWgradBlocks;
  if (n_strips < nb) nb = static_cast<int>(n_strips > 0 ? n_strips : 1);
  const dim3 block(cc / 4, 7), grid(nb, (C + kCC - 1) / kCC);
  hipStream_t s = as_stream(stream);
#define WG_LAUNCH(TX, TD)                                                                                   \
  hipLaunchKernelGGL((dwconv7x7_wgrad_kernel<TX, TD>), grid, block, 0, s, static_cast<const TX*>(x),          \
                     static_cast<const TD*>(dy), ws, H, W, C, tiles_w, n_strips)
  if (x_dtype == APGD_F32 && dy_dtype == APGD_F32) WG_LAUNCH(float, float);
  else if (x_dtype == APGD_F32) WG_LAUNCH(float, uint16_t);
  else if (dy_dtype == APGD_F32) WG_LAUNCH(uint16_t, float);
  else WG_LAUNCH(uint16_t, uint16_t);
#undef WG_LAUNCH
  const int len = 50 * C;
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 63) / 64), dim3(256), 0, s, ws, dw49c, dbias, 49 * C, len, nb);
  return launch_status();
}

int cnx_layernorm_fwd(const void* x, int x_dtype, const float* weight, const float* bias, float eps, void* y,
                      int y_dtype, float* mean, float* rstd, int64_t M, int32_t C, int32_t gelu, void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!x || !weight || !bias || !y) return APGD_ERR_NULL;
  if ((mean == nullptr) != (rstd == nullptr)) return APGD_ERR_ARG;
  if (C % 4 != 0) return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (x_dtype == APGD_F32 && y_dtype == APGD_F32)
    return launch_ln_fwd(static_cast<const float*>(x), weight, bias, eps, static_cast<float*>(y), mean, rstd, M, C, gelu, s);
  if (x_dtype == APGD_F32 && y_dtype == APGD_BF16)
    return launch_ln_fwd(static_cast<const float*>(x), weight, bias, eps, static_cast<uint16_t*>(y), mean, rstd, M, C, gelu, s);
  if (x_dtype == APGD_BF16 && y_dtype == APGD_F32)
    return launch_ln_fwd(static_cast<const uint16_t*>(x), weight, bias, eps, static_cast<float*>(y), mean, rstd, M, C, gelu, s);
  if (x_dtype == APGD_BF16 && y_dtype == APGD_BF16)
    return launch_ln_fwd(static_cast<const uint16_t*>(x), weight, bias, eps, static_cast<uint16_t*>(y), mean, rstd, M, C, gelu, s);
  return APGD_ERR_DTYPE;
}

int64_t cnx_layernorm_bwd_ws_floats(int32_t C) { return static_cast<int64_t>(kLnBwdBlocks) * 2 * C; }

int cnx_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight, const float* bias,
                      const float* mean, const float* rstd, void* dx, int dx_dtype, float* dweight, float* dbias,
                      float* ws, int64_t M, int32_t C, int32_t gelu, void* stream) {
  if (M < 0 || C <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!dy || !x || !weight || !mean || !rstd || !dx) return APGD_ERR_NULL;
  if (gelu && !bias) return APGD_ERR_NULL;
  if (dweight && (!dbias || !ws)) return APGD_ERR_NULL;
  if (C % 4 != 0) return APGD_ERR_ARG;
  if ((dy_dtype | x_dtype | dx_dtype) & ~1) return APGD_ERR_DTYPE;
  hipStream_t s = as_stream(stream);
  float* wsp = dweight ? ws : nullptr;
  int nb = 0, rc = APGD_ERR_DTYPE;
#define LNB(TD, TX, TO)                                                                                          \
  rc = launch_ln_bwd(static_cast<const TD*>(dy), static_cast<const TX*>(x), weight, bias, mean, rstd,            \
                     static_cast<TO*>(dx), wsp, M, C, gelu, &nb, s)
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
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((len + 63) / 64), dim3(256), 0, s, ws, dweight, dbias, C, len, nb);
    return launch_status();
  }
  return APGD_OK;
}

}  // extern "C"
