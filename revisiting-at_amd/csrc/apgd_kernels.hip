// apgd_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels for the APGD inner loop.
//
// Replaces the eager ATen sequences of /root/reference/autopgd_train_clean.py:123-371
// (see include/apgd_hip.h for the per-entry-point line map).  All attack-state kernels are
// HBM-bound element-wise / row-copy work: 16-byte coalesced accesses, 64-wide wavefront
// shuffles for the per-sample reductions, no MFMA, no host synchronisation.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off is REQUIRED: the reference rounds every multiply and add separately.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "apgd_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;        // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kL2Parts = 64;       // partial sums per sample in the L2 path (deterministic order)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- small device helpers
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
__device__ __forceinline__ float clamp01(float t) { return fminf(fmaxf(t, 0.0f), 1.0f); }

template <typename T> struct Elt;
template <> struct Elt<float> {
  static __device__ __forceinline__ float load(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void store(float* p, int64_t i, float v) { p[i] = v; }
};
template <> struct Elt<uint16_t> {  // bf16
  static __device__ __forceinline__ float load(const uint16_t* p, int64_t i) { return bf16_to_f32(p[i]); }
  static __device__ __forceinline__ void store(uint16_t* p, int64_t i, float v) { p[i] = f32_to_bf16_rne(v); }
};
template <> struct Elt<int8_t> {    // gradient signs
  static __device__ __forceinline__ float load(const int8_t* p, int64_t i) { return static_cast<float>(p[i]); }
};
template <> struct Elt<_Float16> {
  static __device__ __forceinline__ float load(const _Float16* p, int64_t i) { return static_cast<float>(p[i]); }
  static __device__ __forceinline__ void store(_Float16* p, int64_t i, float v) { p[i] = static_cast<_Float16>(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, kWave));
  return v;
}
// (value, index) arg-max with "first maximal index" tie-break, as torch's CPU max(1)[1].
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) {
    const float ov = __shfl_xor(v, m, kWave);
    const int oi = __shfl_xor(i, m, kWave);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
// block-wide sum of one float per thread -> every thread gets the total (fixed order).
__device__ __forceinline__ float block_sum(float v, float* lds /* >= kBlock/kWave floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int k = 0; k < kBlock / kWave; ++k) t += lds[k];
  return t;
}

// ---------------------------------------------------------------- a1: prologue
// x_adv = clamp(x,0,1); x_best = x_best_adv = x_adv.  4 B read + up to 12 B written per element.
template <int VEC>
__global__ __launch_bounds__(kBlock) void init_kernel(const float* __restrict__ x, float* __restrict__ xa,
                                                      float* __restrict__ xb, float* __restrict__ xba, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  if constexpr (VEC == 4) {
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; v < n4; v += stride) {
      float4 t = x4[v];
      t.x = clamp01(t.x); t.y = clamp01(t.y); t.z = clamp01(t.z); t.w = clamp01(t.w);
      reinterpret_cast<float4*>(xa)[v] = t;
      if (xb) reinterpret_cast<float4*>(xb)[v] = t;
      if (xba) reinterpret_cast<float4*>(xba)[v] = t;
    }
    for (int64_t e = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < n; e += stride) {
      const float t = clamp01(x[e]);
      xa[e] = t; if (xb) xb[e] = t; if (xba) xba[e] = t;
    }
  } else {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < n; e += stride) {
      const float t = clamp01(x[e]);
      xa[e] = t; if (xb) xb[e] = t; if (xba) xba[e] = t;
    }
  }
}

// ---------------------------------------------------------------- a2: Linf step
// One element of autopgd_train_clean.py:214-226.  Every operation is rounded separately.
__device__ __forceinline__ float linf_elem(float x, float xa, float xo, float g, float st, float eps, float a,
                                           float oma) {
  const float lo = x - eps;                                   // x - eps   (:223)
  const float hi = x + eps;                                   // x + eps
  const float sg = (g > 0.0f) ? st : ((g < 0.0f) ? -st : 0.0f);  // step_size * sign(grad); sign(0)=sign(NaN)=0
  float t = xa + sg;                                          // :221
  t = clamp01(fminf(fmaxf(t, lo), hi));                       // :222-223
  const float grad2 = xa - xo;                                // :214
  float u = (xa + (t - xa) * a) + grad2 * oma;                // :225
  return clamp01(fminf(fmaxf(u, lo), hi));                    // :224-226
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4(const float* p, int64_t v, bool nt) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p) + v;
  const f32x4 t = nt ? __builtin_nontemporal_load(q) : *q;
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4(float* p, int64_t v, float4 r, bool nt) {
  f32x4* q = reinterpret_cast<f32x4*>(p) + v;
  const f32x4 t = {r.x, r.y, r.z, r.w};
  if (nt) __builtin_nontemporal_store(t, q); else *q = t;
}
struct G4 { float x, y, z, w; };
__device__ __forceinline__ G4 ldg4(const float* p, int64_t v, bool nt) {
  const float4 t = ld4(p, v, nt);
  return {t.x, t.y, t.z, t.w};
}
// int8 signs {-1, 0, +1} (4 per dword): what the Linf step consumes of the gradient (:221, torch.sign)
__device__ __forceinline__ G4 ldg4(const int8_t* p, int64_t v, bool nt) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p) + v;
  const uint32_t t = nt ? __builtin_nontemporal_load(q) : *q;
  return {static_cast<float>(static_cast<int8_t>(t & 0xffu)), static_cast<float>(static_cast<int8_t>((t >> 8) & 0xffu)),
          static_cast<float>(static_cast<int8_t>((t >> 16) & 0xffu)), static_cast<float>(static_cast<int8_t>(t >> 24))};
}
__device__ __forceinline__ G4 ldg4(const uint16_t* p, int64_t v, bool nt) {
  const u32x2* q = reinterpret_cast<const u32x2*>(p) + v;     // 4 bf16 = 8 bytes
  const u32x2 t = nt ? __builtin_nontemporal_load(q) : *q;
  return {__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
          __uint_as_float(t.y & 0xffff0000u)};
}

// grid = (blocks_per_sample, B).  Each block walks its sample's float4 range with a
// block-stride loop, U independent float4 per stream in flight per thread.
// iteration 0 (autopgd_train_clean.py:205, 218): x_adv_old == x_adv and a == 1, so grad2 == +0 and
// grad2*(1-a) == +0; u = (x_adv + (x1 - x_adv)*1) + 0 is evaluated without fetching x_adv_old.
// Bit-identical to the general form ((t - xa)*1.0f and "+ 0.0f" are exact for the finite, non-negative
// operands of this path) and one stream lighter: 16 B/element.
__device__ __forceinline__ float linf_elem_first(float x, float xa, float g, float st, float eps) {
  const float lo = x - eps;
  const float hi = x + eps;
  const float sg = (g > 0.0f) ? st : ((g < 0.0f) ? -st : 0.0f);
  float t = xa + sg;
  t = clamp01(fminf(fmaxf(t, lo), hi));
  const float u = xa + (t - xa);
  return clamp01(fminf(fmaxf(u, lo), hi));
}

template <typename GT, bool BF16_OUT>
__global__ __launch_bounds__(kBlock) void linf_step_first_vec4_kernel(
    const float* __restrict__ x, const float* __restrict__ xa, const GT* __restrict__ g,
    const float* __restrict__ step, float* __restrict__ out, uint16_t* __restrict__ out_bf16, int64_t E, float eps) {
  const int64_t b = blockIdx.y;
  const float st = step[b];
  const int64_t row = b * E;
  const int64_t E4 = E >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; v < E4; v += stride) {
    const float4 X = ld4(x + row, v, false), A = ld4(xa + row, v, false);
    const G4 Gd = ldg4(g + row, v, false);
    float4 r;
    r.x = linf_elem_first(X.x, A.x, Gd.x, st, eps);
    r.y = linf_elem_first(X.y, A.y, Gd.y, st, eps);
    r.z = linf_elem_first(X.z, A.z, Gd.z, st, eps);
    r.w = linf_elem_first(X.w, A.w, Gd.w, st, eps);
    st4(out + row, v, r, false);
    if (BF16_OUT) {
      uint2 pk;
      pk.x = static_cast<uint32_t>(f32_to_bf16_rne(r.x)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.y)) << 16);
      pk.y = static_cast<uint32_t>(f32_to_bf16_rne(r.z)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.w)) << 16);
      reinterpret_cast<uint2*>(out_bf16 + row)[v] = pk;
    }
  }
}

template <typename GT, int U, bool BF16_OUT, bool NT>
__global__ __launch_bounds__(kBlock) void linf_step_vec4_kernel(
    const float* __restrict__ x, const float* __restrict__ xa, const float* __restrict__ xo,
    const GT* __restrict__ g, const float* __restrict__ step, float* __restrict__ out,
    uint16_t* __restrict__ out_bf16, int64_t E, float eps, float a, float oma) {
  const int64_t b = blockIdx.y;
  const float st = step[b];                                   // wave-uniform -> scalar load
  const int64_t row = b * E;                                  // E % 4 == 0 on this path
  const float* xr = x + row; const float* xar = xa + row; const float* xor_ = xo + row;
  const GT* gr = g + row; float* outr = out + row;
  uint16_t* obr = BF16_OUT ? out_bf16 + row : nullptr;
  const int64_t E4 = E >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  for (; v + (U - 1) * stride < E4; v += U * stride) {
    float4 X[U], A[U], O[U]; G4 Gd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      X[u] = ld4(xr, v + u * stride, NT); A[u] = ld4(xar, v + u * stride, NT);
      O[u] = ld4(xor_, v + u * stride, NT); Gd[u] = ldg4(gr, v + u * stride, NT);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 r;
      r.x = linf_elem(X[u].x, A[u].x, O[u].x, Gd[u].x, st, eps, a, oma);
      r.y = linf_elem(X[u].y, A[u].y, O[u].y, Gd[u].y, st, eps, a, oma);
      r.z = linf_elem(X[u].z, A[u].z, O[u].z, Gd[u].z, st, eps, a, oma);
      r.w = linf_elem(X[u].w, A[u].w, O[u].w, Gd[u].w, st, eps, a, oma);
      st4(outr, v + u * stride, r, NT);
      if (BF16_OUT) {
        uint2 pk;
        pk.x = static_cast<uint32_t>(f32_to_bf16_rne(r.x)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.y)) << 16);
        pk.y = static_cast<uint32_t>(f32_to_bf16_rne(r.z)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.w)) << 16);
        reinterpret_cast<uint2*>(obr)[v + u * stride] = pk;
      }
    }
  }
  for (; v < E4; v += stride) {
    const float4 X = ld4(xr, v, NT), A = ld4(xar, v, NT), O = ld4(xor_, v, NT);
    const G4 Gd = ldg4(gr, v, NT);
    float4 r;
    r.x = linf_elem(X.x, A.x, O.x, Gd.x, st, eps, a, oma);
    r.y = linf_elem(X.y, A.y, O.y, Gd.y, st, eps, a, oma);
    r.z = linf_elem(X.z, A.z, O.z, Gd.z, st, eps, a, oma);
    r.w = linf_elem(X.w, A.w, O.w, Gd.w, st, eps, a, oma);
    reinterpret_cast<float4*>(outr)[v] = r;
    if (BF16_OUT) {
      uint2 pk;
      pk.x = static_cast<uint32_t>(f32_to_bf16_rne(r.x)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.y)) << 16);
      pk.y = static_cast<uint32_t>(f32_to_bf16_rne(r.z)) | (static_cast<uint32_t>(f32_to_bf16_rne(r.w)) << 16);
      reinterpret_cast<uint2*>(obr)[v] = pk;
    }
  }
}

// Blocked int8 signs (APGD_I8_BLK; E % 1024 == 0).  A wavefront owns one group of 1024 elements of a sample = 256 float4 chunks:
// lane l takes chunks u*64 + l (u = 0..3), so every fp32 load / store is one fully coalesced KiB per wavefront, and finds the
// signs of all four chunks in ONE 16-byte load at byte l*16 of the group (the order cnx_stem_conv_dgrad_sign_blk writes) - a
// KiB per wavefront instead of four 256-byte requests.  FIRST: iteration 0 (x_adv_old is x_adv, a = 1: linf_elem_first).
template <bool FIRST>
__global__ __launch_bounds__(kBlock) void linf_step_i8blk_kernel(
    const float* __restrict__ x, const float* __restrict__ xa, const float* __restrict__ xo, const int8_t* __restrict__ g,
    const float* __restrict__ step, float* __restrict__ out, int64_t E, float eps, float a, float oma) {
  const int64_t b = blockIdx.y;
  const float st = step[b];
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t grp = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + (threadIdx.x / kWave);
  if (grp >= (E >> 10)) return;
  const int64_t row = b * E + (grp << 10);
  const uint4 sg = *reinterpret_cast<const uint4*>(g + row + lane * 16);
  const uint32_t sw[4] = {sg.x, sg.y, sg.z, sg.w};
  float4 X[4], A[4], O[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t v = ((row >> 2) + u * 64 + lane);
    X[u] = ld4(x, v, false);
    A[u] = ld4(xa, v, false);
    if (!FIRST) O[u] = ld4(xo, v, false);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t t = sw[u];
    const float g0 = static_cast<float>(static_cast<int8_t>(t & 0xffu)), g1 = static_cast<float>(static_cast<int8_t>((t >> 8) & 0xffu));
    const float g2 = static_cast<float>(static_cast<int8_t>((t >> 16) & 0xffu)), g3 = static_cast<float>(static_cast<int8_t>(t >> 24));
    float4 r;
    if (FIRST) {
      r.x = linf_elem_first(X[u].x, A[u].x, g0, st, eps); r.y = linf_elem_first(X[u].y, A[u].y, g1, st, eps);
      r.z = linf_elem_first(X[u].z, A[u].z, g2, st, eps); r.w = linf_elem_first(X[u].w, A[u].w, g3, st, eps);
    } else {
      r.x = linf_elem(X[u].x, A[u].x, O[u].x, g0, st, eps, a, oma); r.y = linf_elem(X[u].y, A[u].y, O[u].y, g1, st, eps, a, oma);
      r.z = linf_elem(X[u].z, A[u].z, O[u].z, g2, st, eps, a, oma); r.w = linf_elem(X[u].w, A[u].w, O[u].w, g3, st, eps, a, oma);
    }
    st4(out, (row >> 2) + u * 64 + lane, r, false);
  }
}

// scalar fallback for rows that are not 16-byte tileable (E % 4 != 0 or unaligned views)
template <typename GT>
__global__ __launch_bounds__(kBlock) void linf_step_scalar_kernel(
    const float* __restrict__ x, const float* __restrict__ xa, const float* __restrict__ xo,
    const GT* __restrict__ g, const float* __restrict__ step, float* __restrict__ out,
    uint16_t* __restrict__ out_bf16, int64_t E, float eps, float a, float oma) {
  const int64_t b = blockIdx.y;
  const float st = step[b];
  const int64_t row = b * E;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const float r = linf_elem(x[row + e], xa[row + e], xo[row + e], Elt<GT>::load(g, row + e), st, eps, a, oma);
    out[row + e] = r;
    if (out_bf16) out_bf16[row + e] = f32_to_bf16_rne(r);
  }
}

// ---------------------------------------------------------------- a8: L2 step (4 passes)
__device__ __forceinline__ float parts_total(const float* p) {  // fixed-order sum of kL2Parts partials
  float v = (threadIdx.x < kL2Parts) ? p[threadIdx.x] : 0.0f;
  __shared__ float s_tot;
  if (threadIdx.x < kWave) {
    v = wave_sum(v);
    if (threadIdx.x == 0) s_tot = v;
  }
  __syncthreads();
  return s_tot;
}

// PASS 1: sum g^2 | PASS 2: sum (x1 - x)^2 | PASS 3: sum (u - x)^2 | PASS 4: write projection
template <int PASS>
__global__ __launch_bounds__(kBlock) void l2_step_kernel(
    const float* __restrict__ x, const float* __restrict__ xa, const float* __restrict__ xo,
    const float* __restrict__ g, const float* __restrict__ step, float* __restrict__ out,
    float* __restrict__ ws, int64_t B, int64_t E, float eps, float a, float oma) {
  __shared__ float lds[kBlock / kWave];
  const int64_t b = blockIdx.y;
  const int64_t row = b * E;
  float* ws0 = ws + (0 * B + b) * kL2Parts;
  float* ws1 = ws + (1 * B + b) * kL2Parts;
  float* ws2 = ws + (2 * B + b) * kL2Parts;
  const float st = step[b];
  float ng = 0.f, n1 = 0.f, n2 = 0.f;
  if (PASS >= 2) ng = sqrtf(parts_total(ws0));                // L2_norm(grad)          (:229-230)
  if (PASS >= 3) { __syncthreads(); n1 = sqrtf(parts_total(ws1)); }   // L2_norm(x_adv_1 - x)   (:231-233)
  if (PASS >= 4) { __syncthreads(); n2 = sqrtf(parts_total(ws2)); }   // second projection      (:235-237)
  const float f1 = fminf(eps, n1), f2 = fminf(eps, n2);
  float acc = 0.0f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const float gv = g[row + e];
    if (PASS == 1) { acc += gv * gv; continue; }
    const float xv = x[row + e], av = xa[row + e];
    float x1 = av + st * gv / (ng + 1e-12f);                   // :229-230
    const float d1 = x1 - xv;
    if (PASS == 2) { acc += d1 * d1; continue; }
    x1 = clamp01(xv + d1 / (n1 + 1e-12f) * f1);               // :231-233
    const float grad2 = av - xo[row + e];                      // :214
    const float u = av + (x1 - av) * a + grad2 * oma;          // :234
    const float d2 = u - xv;
    if (PASS == 3) { acc += d2 * d2; continue; }
    out[row + e] = clamp01(xv + d2 / (n2 + 1e-12f) * f2);     // :235-237
  }
  if (PASS < 4) {
    const float tot = block_sum(acc, lds);
    float* dst = PASS == 1 ? ws0 : (PASS == 2 ? ws1 : ws2);
    if (threadIdx.x == 0) dst[blockIdx.x] = tot;               // gridDim.x == kL2Parts
  }
}

// ---------------------------------------------------------------- a3/a4: loss, pred, dlogits
// One wavefront per row; rows are short (n_cls = 1000): 3 cached passes over the row.
template <typename T>
__global__ __launch_bounds__(kBlock) void ce_pred_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ y_hard,
    const float* __restrict__ y_soft, float* __restrict__ loss, uint8_t* __restrict__ pred,
    T* __restrict__ dlogits, int64_t B, int64_t C) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  if (b >= B) return;
  const T* z = logits + b * ld;
  // pass 1: max / argmax of the logits (first maximal index)
  float m = -INFINITY; int am = 0x7fffffff;
  for (int64_t c = lane; c < C; c += kWave) {
    const float v = Elt<T>::load(z, c);
    if (v > m) { m = v; am = static_cast<int>(c); }
  }
  wave_argmax(m, am);
  // pass 2: sum exp(z - m); soft labels: sum y, sum y*(z-m), argmax y
  float se = 0.f, sy = 0.f, syz = 0.f, ym = -INFINITY; int yam = 0x7fffffff;
  const float* ys = y_soft ? y_soft + b * C : nullptr;
  for (int64_t c = lane; c < C; c += kWave) {
    const float d = Elt<T>::load(z, c) - m;
    se += expf(d);
    if (ys) {
      const float yv = ys[c];
      sy += yv; syz += yv * d;
      if (yv > ym) { ym = yv; yam = static_cast<int>(c); }
    }
  }
  se = wave_sum(se);
  const float lse = logf(se);
  int64_t yh = 0; float l;
  if (ys) {
    sy = wave_sum(sy); syz = wave_sum(syz); wave_argmax(ym, yam);
    l = -(syz - sy * lse);                     // -sum_c y_c * ((z_c - m) - lse)
  } else {
    yh = y_hard[b];
    if (yh < 0 || yh >= C) {                   // torch raises a device assert here; we poison the sample
      l = NAN; yh = -1;
    } else {
      const float zy = Elt<T>::load(z, yh) - m;
      l = -(zy - lse);                         // -log_softmax(z)[y]
    }
    sy = 1.0f;
  }
  if (lane == 0) {
    loss[b] = l;
    pred[b] = ys ? (am == yam) : (static_cast<int64_t>(am) == yh);
  }
  // pass 3: d(sum loss)/dz = softmax * sum(y) - y
  if (dlogits) {
    T* dz = dlogits + b * ld;
    const float inv = 1.0f / se;
    for (int64_t c = lane; c < C; c += kWave) {
      const float p = expf(Elt<T>::load(z, c) - m) * inv;
      const float t = ys ? ys[c] : ((c == yh) ? 1.0f : 0.0f);
      Elt<T>::store(dz, c, p * sy - t);
    }
  }
}

// ---------------------------------------------------------------- a18: DLR loss (dlr_loss, :99-104)
// Sorted-order semantics of torch.sort (stable, ascending): among equal values the HIGHER index sorts
// later, so "descending" order here is (value desc, index desc).
struct Top3 { float v[3]; int i[3]; };
__device__ __forceinline__ bool dlr_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia > ib); }
__device__ __forceinline__ void top3_insert(Top3& t, float v, int i) {
  if (dlr_before(v, i, t.v[0], t.i[0])) { t.v[2] = t.v[1]; t.i[2] = t.i[1]; t.v[1] = t.v[0]; t.i[1] = t.i[0]; t.v[0] = v; t.i[0] = i; }
  else if (dlr_before(v, i, t.v[1], t.i[1])) { t.v[2] = t.v[1]; t.i[2] = t.i[1]; t.v[1] = v; t.i[1] = i; }
  else if (dlr_before(v, i, t.v[2], t.i[2])) { t.v[2] = v; t.i[2] = i; }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void dlr_pred_kernel(const T* __restrict__ logits, int64_t ld,
                                                          const int64_t* __restrict__ y_hard, float* __restrict__ loss,
                                                          uint8_t* __restrict__ pred, T* __restrict__ dlogits, int64_t B,
                                                          int64_t C) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  if (b >= B) return;
  const T* z = logits + b * ld;
  Top3 t; float m = -INFINITY; int am = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 3; ++k) { t.v[k] = -INFINITY; t.i[k] = -1 - k; }
  for (int64_t c = lane; c < C; c += kWave) {
    const float v = Elt<T>::load(z, c);
    top3_insert(t, v, static_cast<int>(c));
    if (v > m) { m = v; am = static_cast<int>(c); }
  }
#pragma unroll
  for (int mk = kWave / 2; mk > 0; mk >>= 1) {
    float ov[3]; int oi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ov[k] = __shfl_xor(t.v[k], mk, kWave); oi[k] = __shfl_xor(t.i[k], mk, kWave); }
#pragma unroll
    for (int k = 0; k < 3; ++k) top3_insert(t, ov[k], oi[k]);
  }
  wave_argmax(m, am);                                          // first maximal index: logits.max(1)[1] (:197, 294)
  const int64_t yh = y_hard[b];
  const bool ok = yh >= 0 && yh < C;
  const float zy = ok ? Elt<T>::load(z, yh) : NAN;
  const float ind = (static_cast<int64_t>(t.i[0]) == yh) ? 1.0f : 0.0f;      // ind_sorted[:, -1] == y  (:101)
  const float num = (zy - t.v[1] * ind) - t.v[0] * (1.0f - ind);            // :103
  const float den = (t.v[0] - t.v[2]) + 1e-12f;                              // :104
  if (lane == 0) { loss[b] = -num / den; pred[b] = static_cast<int64_t>(am) == yh; }
  if (dlogits) {
    // loss = -N/D: dN = +1 at y, -1 at (ind ? i2 : i1); dD = +1 at i1, -1 at i3
    T* dz = dlogits + b * ld;
    const float inv = 1.0f / den, nd2 = num * inv * inv;
    const int isub = ind != 0.0f ? t.i[1] : t.i[0];
    for (int64_t c = lane; c < C; c += kWave) {
      float gsum = 0.0f;
      if (c == yh) gsum -= inv;
      if (c == isub) gsum += inv;
      if (c == t.i[0]) gsum += nd2;
      if (c == t.i[2]) gsum -= nd2;
      Elt<T>::store(dz, c, gsum);
    }
  }
}

// ---------------------------------------------------------------- a18: targeted DLR (dlr_loss_targeted, :106-111)
//   loss = -(z_y - z_t) / (z_(1) - 0.5 (z_(3) + z_(4)) + 1e-12),  z_(k) = k-th largest logit.
// Top-4 per row with the same (value desc, index desc) order as above; gradient:
//   d loss = -(e_y - e_t)/D + N/D^2 (e_i1 - 0.5 e_i3 - 0.5 e_i4).
struct Top4 { float v[4]; int i[4]; };
__device__ __forceinline__ void top4_insert(Top4& t, float v, int i) {
  if (dlr_before(v, i, t.v[3], t.i[3])) {
    t.v[3] = v; t.i[3] = i;
#pragma unroll
    for (int k = 3; k > 0; --k) {
      if (dlr_before(t.v[k], t.i[k], t.v[k - 1], t.i[k - 1])) {
        const float fv = t.v[k]; t.v[k] = t.v[k - 1]; t.v[k - 1] = fv;
        const int fi = t.i[k]; t.i[k] = t.i[k - 1]; t.i[k - 1] = fi;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void dlr_targeted_pred_kernel(const T* __restrict__ logits, int64_t ld,
                                                                   const int64_t* __restrict__ y_hard,
                                                                   const int64_t* __restrict__ y_target,
                                                                   float* __restrict__ loss, uint8_t* __restrict__ pred,
                                                                   T* __restrict__ dlogits, int64_t B, int64_t C) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
  if (b >= B) return;
  const T* z = logits + b * ld;
  Top4 t; float m = -INFINITY; int am = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k) { t.v[k] = -INFINITY; t.i[k] = -1 - k; }
  for (int64_t c = lane; c < C; c += kWave) {
    const float v = Elt<T>::load(z, c);
    top4_insert(t, v, static_cast<int>(c));
    if (v > m) { m = v; am = static_cast<int>(c); }
  }
#pragma unroll
  for (int mk = kWave / 2; mk > 0; mk >>= 1) {
    float ov[4]; int oi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ov[k] = __shfl_xor(t.v[k], mk, kWave); oi[k] = __shfl_xor(t.i[k], mk, kWave); }
#pragma unroll
    for (int k = 0; k < 4; ++k) top4_insert(t, ov[k], oi[k]);
  }
  wave_argmax(m, am);
  const int64_t yh = y_hard[b], yt = y_target[b];
  const bool ok = yh >= 0 && yh < C && yt >= 0 && yt < C;
  const float zy = ok ? Elt<T>::load(z, yh) : NAN;
  const float zt = ok ? Elt<T>::load(z, yt) : NAN;
  const float num = zy - zt;                                                 // :110
  const float den = (t.v[0] - 0.5f * (t.v[2] + t.v[3])) + 1e-12f;            // :110-111
  if (lane == 0) { loss[b] = -num / den; pred[b] = static_cast<int64_t>(am) == yh; }
  if (dlogits) {
    T* dz = dlogits + b * ld;
    const float inv = 1.0f / den, nd2 = num * inv * inv;
    for (int64_t c = lane; c < C; c += kWave) {
      float gsum = 0.0f;
      if (c == yh) gsum -= inv;
      if (c == yt) gsum += inv;
      if (c == t.i[0]) gsum += nd2;
      if (c == t.i[2]) gsum -= 0.5f * nd2;
      if (c == t.i[3]) gsum -= 0.5f * nd2;
      Elt<T>::store(dz, c, gsum);
    }
  }
}

// ---------------------------------------------------------------- a4-a6: per-sample state machine
__global__ __launch_bounds__(kBlock) void state_update_kernel(
    const float* __restrict__ loss, const uint8_t* __restrict__ pred, uint8_t* __restrict__ acc,
    float* __restrict__ loss_best, float* __restrict__ loss_best_last, float* __restrict__ reduced_last,
    float* __restrict__ step_size, float* __restrict__ loss_steps, uint8_t* __restrict__ flags,
    int64_t B, int K, int i, int do_check, int k, float thr) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  const float y1 = loss[b];                                    // :319
  loss_steps[static_cast<int64_t>(i) * B + b] = y1;            // :320
  const uint8_t p = pred[b];
  acc[b] = acc[b] & p;                                         // :296  min(acc, pred)
  uint32_t f = p ? 0u : APGD_FLAG_MISCLS;                      // :301
  float lb = loss_best[b];
  if (y1 > lb) { f |= APGD_FLAG_NEW_BEST; lb = y1; loss_best[b] = lb; }   // :321, 324
  if (do_check) {                                              // :329  counter3 == k
    float t = 0.0f;                                            // check_oscillation :116-121
    for (int c = 0; c < k; ++c) {
      const int j = i - c;
      int jm = j - 1; if (jm < 0) jm += K;                     // python negative index wrap
      const float cur = (j == i) ? y1 : loss_steps[static_cast<int64_t>(j) * B + b];
      const float prv = (jm == i) ? y1 : loss_steps[static_cast<int64_t>(jm) * B + b];
      t += (cur > prv) ? 1.0f : 0.0f;
    }
    const float osc = (t <= thr) ? 1.0f : 0.0f;                // :121
    const float noimp = (1.0f - reduced_last[b]) * ((loss_best_last[b] >= lb) ? 1.0f : 0.0f);  // :333-334
    const float fl = fmaxf(osc, noimp);                        // :335-336
    reduced_last[b] = fl;                                      // :337
    loss_best_last[b] = lb;                                    // :338
    if (fl > 0.0f) { step_size[b] = step_size[b] / 2.0f; f |= APGD_FLAG_HALVE; }   // :341-342
  }
  flags[b] = static_cast<uint8_t>(f);
}

// ---------------------------------------------------------------- a4-a6: row moves
// grid = (blocks_per_sample, B).  Rows whose flag byte needs no copy cost one byte read.
__device__ __forceinline__ void copy_bytes(void* __restrict__ dst, const void* __restrict__ src, int64_t nbytes,
                                           bool vec16, int64_t first, int64_t stride) {
  if (vec16) {
    const uint4* s = static_cast<const uint4*>(src); uint4* d = static_cast<uint4*>(dst);
    const int64_t n = nbytes >> 4;
    for (int64_t v = first; v < n; v += stride) d[v] = s[v];
  } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | static_cast<uintptr_t>(nbytes)) & 1) == 0) {
    const uint16_t* s = static_cast<const uint16_t*>(src); uint16_t* d = static_cast<uint16_t*>(dst);
    const int64_t n = nbytes >> 1;
    for (int64_t v = first; v < n; v += stride) d[v] = s[v];
  } else {                                // int8 sign rows of odd length (E odd): rows start on odd addresses, copy bytes
    const uint8_t* s = static_cast<const uint8_t*>(src); uint8_t* d = static_cast<uint8_t*>(dst);
    for (int64_t v = first; v < nbytes; v += stride) d[v] = s[v];
  }
}

__global__ __launch_bounds__(kBlock) void track_rows_kernel(
    const uint8_t* __restrict__ flags, float* x_adv, uint8_t* grad, float* x_best, uint8_t* grad_best,
    float* x_best_adv, int grad_elt, int64_t E, int final_iter, int vec16) {
  const int64_t b = blockIdx.y;
  const uint32_t f = flags[b];
  const bool nb = f & APGD_FLAG_NEW_BEST, mc = f & APGD_FLAG_MISCLS;
  const bool rs = (f & APGD_FLAG_HALVE) && !nb && !final_iter;   // NEW_BEST makes the restore a no-op
  if (!(nb || mc || rs)) return;
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t xbytes = E * 4, gbytes = E * grad_elt;
  float* xa = x_adv + b * E; float* xb = x_best + b * E; float* xba = x_best_adv + b * E;
  if (nb && mc && vec16) {            // one read of x_adv feeds both destinations (:304, :322)
    const uint4* s = reinterpret_cast<const uint4*>(xa);
    uint4* d0 = reinterpret_cast<uint4*>(xb); uint4* d1 = reinterpret_cast<uint4*>(xba);
    for (int64_t v = first; v < (xbytes >> 4); v += stride) { const uint4 t = s[v]; d0[v] = t; d1[v] = t; }
  } else {
    if (nb) copy_bytes(xb, xa, xbytes, vec16, first, stride);                 // :322
    if (mc) copy_bytes(xba, xa, xbytes, vec16, first, stride);                // :304
  }
  if (grad && !final_iter) {
    uint8_t* gr = grad + b * gbytes; uint8_t* gb = grad_best + b * gbytes;
    if (nb) copy_bytes(gb, gr, gbytes, vec16, first, stride);                 // :323
    if (rs) copy_bytes(gr, gb, gbytes, vec16, first, stride);                 // :346
  }
  if (rs) copy_bytes(xa, xb, xbytes, vec16, first, stride);                   // :345
}

// ---------------------------------------------------------------- invariants (utils_eval.py:67-81)
__global__ __launch_bounds__(kBlock) void check_imgs_kernel(const float* __restrict__ adv, const float* __restrict__ x,
                                                            float* __restrict__ out, int64_t E) {
  __shared__ float s[3][kBlock / kWave];
  const int64_t row = static_cast<int64_t>(blockIdx.x) * E;
  float dm = 0.0f, lo = INFINITY, hi = -INFINITY;
  for (int64_t e = threadIdx.x; e < E; e += kBlock) {
    const float a = adv[row + e];
    dm = fmaxf(dm, fabsf(a - x[row + e])); lo = fminf(lo, a); hi = fmaxf(hi, a);
    if (a != a) hi = a;   // NaN poisons the max so the host check trips
  }
  dm = wave_max(dm); lo = wave_min(lo);
  const bool any_nan = __any(hi != hi);
  hi = wave_max(hi);
  if (any_nan) hi = NAN;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) { s[0][w] = dm; s[1][w] = lo; s[2][w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool nan = false;
    for (int k = 0; k < kBlock / kWave; ++k) {
      dm = fmaxf(dm, s[0][k]); lo = fminf(lo, s[1][k]);
      if (s[2][k] != s[2][k]) nan = true; else hi = fmaxf(hi, s[2][k]);
    }
    out[blockIdx.x * 3 + 0] = dm; out[blockIdx.x * 3 + 1] = lo; out[blockIdx.x * 3 + 2] = nan ? NAN : hi;
  }
}

inline int blocks_for(int64_t work_items, int64_t per_block, int64_t cap) {
  int64_t n = (work_items + per_block - 1) / per_block;
  if (n < 1) n = 1;
  if (n > cap) n = cap;
  return static_cast<int>(n);
}

// ---------------------------------------------------------------- a2 + a4-a6 in one pass
// The Linf step of iteration i + 1 reads x_adv(i) and grad(i) anyway: it also performs the row moves iteration i decided on
// (apgd_state_update's flag byte; :304, 322-323, 345-346) instead of leaving them to a pass of their own:
//   NEW_BEST : x_best <- x_adv, grad_best <- grad                               (4 + sizeof(GT) more bytes written per element)
//   MISCLS   : x_best_adv <- x_adv                                              (4 more)
//   HALVE (and not NEW_BEST - then the restore is a no-op): the step runs from x_best / grad_best instead of x_adv / grad,
//              and x_adv <- x_best so that the buffer is the restored iterate when the rotation makes it x_adv_old (:215, 345);
//              grad is not rewritten: the next backward replaces it (:277-283) and nothing else reads it.
// The flags are per sample = per blockIdx.y: wave-uniform branches, the element math is linf_elem / linf_elem_first unchanged.
// FIRST (iteration 0; flags == NULL): x_adv_old is x_adv, a == 1, and the prologue's clones are made here - x_best = x_best_adv =
// x_adv (:142-143), grad_best = grad (:189) - so that apgd_init_f32 writes x_adv only.
template <typename GT> struct Raw4;
template <> struct Raw4<float> {
  typedef f32x4 T;
  static __device__ __forceinline__ G4 cvt(T t) { return {t.x, t.y, t.z, t.w}; }
};
template <> struct Raw4<uint16_t> {
  typedef u32x2 T;
  static __device__ __forceinline__ G4 cvt(T t) {
    return {__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
  }
};
template <> struct Raw4<int8_t> {
  typedef uint32_t T;
  static __device__ __forceinline__ G4 cvt(T t) {
    return {static_cast<float>(static_cast<int8_t>(t & 0xffu)), static_cast<float>(static_cast<int8_t>((t >> 8) & 0xffu)),
            static_cast<float>(static_cast<int8_t>((t >> 16) & 0xffu)), static_cast<float>(static_cast<int8_t>(t >> 24))};
  }
};

template <typename GT, bool FIRST>
__global__ __launch_bounds__(kBlock) void linf_step_track_vec4_kernel(
    const float* __restrict__ x, float* xa, const float* xo, const GT* g, const float* __restrict__ step,
    float* __restrict__ out, const uint8_t* __restrict__ flags, float* xb, GT* gb, float* xba,
    int64_t E, float eps, float a, float oma) {
  typedef typename Raw4<GT>::T RT;
  const int64_t b = blockIdx.y;
  const float st = step[b];
  const uint32_t f = FIRST ? (APGD_FLAG_NEW_BEST | APGD_FLAG_MISCLS) : flags[b];
  const bool nb = f & APGD_FLAG_NEW_BEST, mc = f & APGD_FLAG_MISCLS;
  const bool rs = (f & APGD_FLAG_HALVE) && !nb;
  const int64_t row = b * E;
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;      // one float4 per stream per thread
  if (v >= (E >> 2)) return;
  const f32x4 X = reinterpret_cast<const f32x4*>(x + row)[v];
  f32x4 A = reinterpret_cast<const f32x4*>(xa + row)[v];
  f32x4 O;
  if (!FIRST) O = reinterpret_cast<const f32x4*>(xo + row)[v];
  RT Gr = reinterpret_cast<const RT*>(g + row)[v];
  if (mc) reinterpret_cast<f32x4*>(xba + row)[v] = A;                             // :304 (the iterate as the model saw it)
  if (nb) {                                                                       // :322-323
    reinterpret_cast<f32x4*>(xb + row)[v] = A;
    reinterpret_cast<RT*>(gb + row)[v] = Gr;
  } else if (rs) {                                                                // :345-346
    A = reinterpret_cast<const f32x4*>(xb + row)[v];
    Gr = reinterpret_cast<const RT*>(gb + row)[v];
    reinterpret_cast<f32x4*>(xa + row)[v] = A;
  }
  const G4 Gd = Raw4<GT>::cvt(Gr);
  f32x4 r;
  if (FIRST) {
    r.x = linf_elem_first(X.x, A.x, Gd.x, st, eps); r.y = linf_elem_first(X.y, A.y, Gd.y, st, eps);
    r.z = linf_elem_first(X.z, A.z, Gd.z, st, eps); r.w = linf_elem_first(X.w, A.w, Gd.w, st, eps);
  } else {
    r.x = linf_elem(X.x, A.x, O.x, Gd.x, st, eps, a, oma); r.y = linf_elem(X.y, A.y, O.y, Gd.y, st, eps, a, oma);
    r.z = linf_elem(X.z, A.z, O.z, Gd.z, st, eps, a, oma); r.w = linf_elem(X.w, A.w, O.w, Gd.w, st, eps, a, oma);
  }
  reinterpret_cast<f32x4*>(out + row)[v] = r;
}

// any E / alignment: one element per thread
template <typename GT, bool FIRST>
__global__ __launch_bounds__(kBlock) void linf_step_track_scalar_kernel(
    const float* __restrict__ x, float* xa, const float* xo, const GT* g, const float* __restrict__ step,
    float* __restrict__ out, const uint8_t* __restrict__ flags, float* xb, GT* gb, float* xba,
    int64_t E, float eps, float a, float oma) {
  const int64_t b = blockIdx.y;
  const float st = step[b];
  const uint32_t f = FIRST ? (APGD_FLAG_NEW_BEST | APGD_FLAG_MISCLS) : flags[b];
  const bool nb = f & APGD_FLAG_NEW_BEST, mc = f & APGD_FLAG_MISCLS;
  const bool rs = (f & APGD_FLAG_HALVE) && !nb;
  const int64_t row = b * E;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const float X = x[row + e];
    float A = xa[row + e];
    GT Gr = g[row + e];
    if (mc) xba[row + e] = A;
    if (nb) { xb[row + e] = A; gb[row + e] = Gr; }
    else if (rs) { A = xb[row + e]; Gr = gb[row + e]; xa[row + e] = A; }
    const float G = Elt<GT>::load(&Gr, 0);
    out[row + e] = FIRST ? linf_elem_first(X, A, G, st, eps) : linf_elem(X, A, xo[row + e], G, st, eps, a, oma);
  }
}

template <typename GT>
int linf_track_dispatch(const float* x, float* xa, const float* xo, const GT* g, const float* step, float* out,
                        const uint8_t* flags, float* xb, GT* gb, float* xba, int64_t B, int64_t E, float eps, float a,
                        bool vec, hipStream_t s) {
  const float oma = static_cast<float>(1.0 - static_cast<double>(a));   // python: (1 - a) in double, then fp32
  const bool first = flags == nullptr;
  if (vec) {
    const dim3 grid(blocks_for(E / 4, kBlock, 65535), static_cast<unsigned>(B));
    if (static_cast<int64_t>(grid.x) * kBlock < E / 4) return APGD_ERR_SIZE;
    if (first)
      hipLaunchKernelGGL((linf_step_track_vec4_kernel<GT, true>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out, flags, xb, gb, xba, E, eps, a, oma);
    else
      hipLaunchKernelGGL((linf_step_track_vec4_kernel<GT, false>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out, flags, xb, gb, xba, E, eps, a, oma);
  } else {
    const dim3 grid(blocks_for(E, kBlock * 4, 64), static_cast<unsigned>(B));
    if (first)
      hipLaunchKernelGGL((linf_step_track_scalar_kernel<GT, true>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out, flags, xb, gb, xba, E, eps, a, oma);
    else
      hipLaunchKernelGGL((linf_step_track_scalar_kernel<GT, false>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out, flags, xb, gb, xba, E, eps, a, oma);
  }
  return launch_status();
}

template <typename GT, int U, bool NT>
int launch_linf_vec4(const float* x, const float* xa, const float* xo, const GT* g, const float* step, float* out,
                     uint16_t* ob, int64_t B, int64_t E, float eps, float a, int bps, hipStream_t s) {
  const dim3 grid(bps, static_cast<unsigned>(B));
  const float oma = static_cast<float>(1.0 - static_cast<double>(a));   // python: (1 - a) in double, then fp32
  if (ob)
    hipLaunchKernelGGL((linf_step_vec4_kernel<GT, U, true, NT>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out, ob,
                       E, eps, a, oma);
  else
    hipLaunchKernelGGL((linf_step_vec4_kernel<GT, U, false, NT>), grid, dim3(kBlock), 0, s, x, xa, xo, g, step, out,
                       ob, E, eps, a, oma);
  return launch_status();
}

}  // namespace

// ======================================================================= C ABI
extern "C" {

int apgd_hip_version(void) { return APGD_HIP_VERSION; }

const char* apgd_hip_strerror(int code) {
  switch (code) {
    case APGD_OK: return "ok";
    case APGD_ERR_NULL: return "required pointer is NULL";
    case APGD_ERR_SIZE: return "negative or inconsistent size";
    case APGD_ERR_DTYPE: return "unknown dtype code";
    case APGD_ERR_ARG: return "invalid argument";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
  }
}

int apgd_init_f32(const float* x, float* x_adv, float* x_best, float* x_best_adv, int64_t n, void* stream) {
  if (n < 0) return APGD_ERR_SIZE;
  if (n == 0) return APGD_OK;
  if (!x || !x_adv) return APGD_ERR_NULL;
  const bool v = aligned16(x) && aligned16(x_adv) && aligned16(x_best) && aligned16(x_best_adv);
  const int grid = blocks_for(n, static_cast<int64_t>(kBlock) * 4 * 4, 4096);
  if (v) hipLaunchKernelGGL(init_kernel<4>, dim3(grid), dim3(kBlock), 0, as_stream(stream), x, x_adv, x_best, x_best_adv, n);
  else hipLaunchKernelGGL(init_kernel<1>, dim3(grid), dim3(kBlock), 0, as_stream(stream), x, x_adv, x_best, x_best_adv, n);
  return launch_status();
}

}  // extern "C"

namespace {

template <typename GT>
int linf_dispatch(const float* x, const float* x_adv, const float* x_adv_old, const GT* g, const float* step_size,
                  float* out, uint16_t* out_bf16, int64_t B, int64_t E, float eps, float a, int32_t blocks_per_sample,
                  int32_t unroll, int32_t nontemporal, bool vec, hipStream_t s) {
  if (!vec) {
    const dim3 grid(blocks_for(E, kBlock * 4, 64), static_cast<unsigned>(B));
    const float oma = static_cast<float>(1.0 - static_cast<double>(a));
    hipLaunchKernelGGL(linf_step_scalar_kernel<GT>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, g, step_size, out,
                       out_bf16, E, eps, a, oma);
    return launch_status();
  }
  const int U = unroll > 0 ? unroll : 1;
  const int64_t E4 = E / 4;
  int bps = blocks_per_sample;
  if (bps <= 0) {
    // measured on MI355X (tools/k1_sweep.py, B=256 x 3x224x224): one float4 per stream per thread and
    // as many workgroups as that takes (147 x 256 = 37632 here) beats block-stride loops by ~8 %
    bps = blocks_for(E4, static_cast<int64_t>(kBlock) * U, 65535);
  }
  if (x_adv_old == x_adv && a == 1.0f && blocks_per_sample <= 0 && unroll <= 0 && !nontemporal) {
    const dim3 grid(bps, static_cast<unsigned>(B));
    if (out_bf16)
      hipLaunchKernelGGL((linf_step_first_vec4_kernel<GT, true>), grid, dim3(kBlock), 0, s, x, x_adv, g, step_size, out,
                         out_bf16, E, eps);
    else
      hipLaunchKernelGGL((linf_step_first_vec4_kernel<GT, false>), grid, dim3(kBlock), 0, s, x, x_adv, g, step_size, out,
                         out_bf16, E, eps);
    return launch_status();
  }
#define APGD_DISPATCH_U(UU)                                                                                              \
  if (U == UU) {                                                                                                         \
    if (nontemporal)                                                                                                     \
      return launch_linf_vec4<GT, UU, true>(x, x_adv, x_adv_old, g, step_size, out, out_bf16, B, E, eps, a, bps, s);    \
    return launch_linf_vec4<GT, UU, false>(x, x_adv, x_adv_old, g, step_size, out, out_bf16, B, E, eps, a, bps, s);     \
  }
  APGD_DISPATCH_U(1) APGD_DISPATCH_U(2) APGD_DISPATCH_U(4)
#undef APGD_DISPATCH_U
  return APGD_ERR_ARG;
}
}  // namespace

extern "C" {

// Tunable form (used by bench/microbench sweeps): blocks_per_sample <= 0 and unroll <= 0 pick defaults.
int apgd_linf_step_f32_ex(const float* x, const float* x_adv, const float* x_adv_old, const void* grad,
                          int grad_dtype, const float* step_size, float* out, uint16_t* out_bf16, int64_t B,
                          int64_t E, float eps, float a, int32_t blocks_per_sample, int32_t unroll,
                          int32_t nontemporal, void* stream) {
  if (B < 0 || E < 0) return APGD_ERR_SIZE;
  if (B == 0 || E == 0) return APGD_OK;
  if (!x || !x_adv || !x_adv_old || !grad || !step_size || !out) return APGD_ERR_NULL;
  if (grad_dtype != APGD_F32 && grad_dtype != APGD_BF16 && grad_dtype != APGD_I8 && grad_dtype != APGD_I8_BLK) return APGD_ERR_DTYPE;
  if (B > 65535) return APGD_ERR_SIZE;
  if (out == x || out == x_adv || out == x_adv_old || out == grad) return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (grad_dtype == APGD_I8_BLK) {
    // blocked signs: whole 1024-element groups, 16-byte aligned streams, no bf16 copy of the result
    if (E % 1024 != 0 || out_bf16 || !(aligned16(x) && aligned16(x_adv) && aligned16(x_adv_old) && aligned16(out) && aligned16(grad)))
      return APGD_ERR_ARG;
    const dim3 grid(static_cast<unsigned>(((E >> 10) + kBlock / kWave - 1) / (kBlock / kWave)), static_cast<unsigned>(B));
    const float oma = static_cast<float>(1.0 - static_cast<double>(a));
    const auto* gs = static_cast<const int8_t*>(grad);
    if (x_adv_old == x_adv && a == 1.0f)
      hipLaunchKernelGGL(linf_step_i8blk_kernel<true>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, gs, step_size, out, E, eps, a, oma);
    else
      hipLaunchKernelGGL(linf_step_i8blk_kernel<false>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, gs, step_size, out, E, eps, a, oma);
    return launch_status();
  }
  const int galign = grad_dtype == APGD_F32 ? 16 : (grad_dtype == APGD_BF16 ? 8 : 4);
  const bool vec = (E % 4 == 0) && aligned16(x) && aligned16(x_adv) && aligned16(x_adv_old) && aligned16(out) &&
                   (reinterpret_cast<uintptr_t>(grad) % galign == 0) &&
                   (!out_bf16 || reinterpret_cast<uintptr_t>(out_bf16) % 8 == 0);
  if (grad_dtype == APGD_F32)
    return linf_dispatch(x, x_adv, x_adv_old, static_cast<const float*>(grad), step_size, out, out_bf16, B, E, eps, a,
                         blocks_per_sample, unroll, nontemporal, vec, s);
  if (grad_dtype == APGD_BF16)
    return linf_dispatch(x, x_adv, x_adv_old, static_cast<const uint16_t*>(grad), step_size, out, out_bf16, B, E, eps, a,
                         blocks_per_sample, unroll, nontemporal, vec, s);
  return linf_dispatch(x, x_adv, x_adv_old, static_cast<const int8_t*>(grad), step_size, out, out_bf16, B, E, eps, a,
                       blocks_per_sample, unroll, nontemporal, vec, s);
}

int apgd_linf_step_f32(const float* x, const float* x_adv, const float* x_adv_old, const void* grad, int grad_dtype,
                       const float* step_size, float* out, uint16_t* out_bf16, int64_t B, int64_t E, float eps,
                       float a, void* stream) {
  // (launch-shape experiments go through apgd_linf_step_f32_ex: tools/k1_sweep.py; the defaults 0, 0, 0 are the measured best)
  return apgd_linf_step_f32_ex(x, x_adv, x_adv_old, grad, grad_dtype, step_size, out, out_bf16, B, E, eps, a,
                               0, 0, 0, stream);
}

int apgd_linf_step_track_f32(const float* x, float* x_adv, const float* x_adv_old, const void* grad, int grad_dtype,
                             const float* step_size, float* out, const uint8_t* flags, float* x_best, void* grad_best,
                             float* x_best_adv, int64_t B, int64_t E, float eps, float a, void* stream) {
  if (B < 0 || E < 0) return APGD_ERR_SIZE;
  if (B == 0 || E == 0) return APGD_OK;
  if (!x || !x_adv || !x_adv_old || !grad || !step_size || !out || !x_best || !grad_best || !x_best_adv) return APGD_ERR_NULL;
  if (grad_dtype != APGD_F32 && grad_dtype != APGD_BF16 && grad_dtype != APGD_I8) return APGD_ERR_DTYPE;
  if (B > 65535) return APGD_ERR_SIZE;
  if (!flags && (x_adv_old != x_adv || a != 1.0f)) return APGD_ERR_ARG;      // the first-iteration form is iteration 0's
  if (out == x || out == x_adv || out == x_adv_old || out == grad || out == x_best || out == x_best_adv || out == grad_best ||
      x_best == x_best_adv || x_best == x_adv || x_best_adv == x_adv || grad_best == grad || (flags && x_adv_old == x_adv))
    return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int galign = grad_dtype == APGD_F32 ? 16 : (grad_dtype == APGD_BF16 ? 8 : 4);
  const bool vec = (E % 4 == 0) && aligned16(x) && aligned16(x_adv) && aligned16(x_adv_old) && aligned16(out) &&
                   aligned16(x_best) && aligned16(x_best_adv) && (reinterpret_cast<uintptr_t>(grad) % galign == 0) &&
                   (reinterpret_cast<uintptr_t>(grad_best) % galign == 0);
  if (grad_dtype == APGD_F32)
    return linf_track_dispatch(x, x_adv, x_adv_old, static_cast<const float*>(grad), step_size, out, flags, x_best,
                               static_cast<float*>(grad_best), x_best_adv, B, E, eps, a, vec, s);
  if (grad_dtype == APGD_BF16)
    return linf_track_dispatch(x, x_adv, x_adv_old, static_cast<const uint16_t*>(grad), step_size, out, flags, x_best,
                               static_cast<uint16_t*>(grad_best), x_best_adv, B, E, eps, a, vec, s);
  return linf_track_dispatch(x, x_adv, x_adv_old, static_cast<const int8_t*>(grad), step_size, out, flags, x_best,
                             static_cast<int8_t*>(grad_best), x_best_adv, B, E, eps, a, vec, s);
}

int apgd_l2_parts(void) { return kL2Parts; }

int apgd_l2_step_f32(const float* x, const float* x_adv, const float* x_adv_old, const float* grad,
                     const float* step_size, float* out, float* ws, int64_t B, int64_t E, float eps, float a,
                     void* stream) {
  if (B < 0 || E < 0) return APGD_ERR_SIZE;
  if (B == 0 || E == 0) return APGD_OK;
  if (!x || !x_adv || !x_adv_old || !grad || !step_size || !out || !ws) return APGD_ERR_NULL;
  if (B > 65535) return APGD_ERR_SIZE;
  if (out == x || out == x_adv || out == x_adv_old || out == grad) return APGD_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const dim3 grid(kL2Parts, static_cast<unsigned>(B));
  const float oma = static_cast<float>(1.0 - static_cast<double>(a));
  hipLaunchKernelGGL(l2_step_kernel<1>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, grad, step_size, out, ws, B, E, eps, a, oma);
  hipLaunchKernelGGL(l2_step_kernel<2>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, grad, step_size, out, ws, B, E, eps, a, oma);
  hipLaunchKernelGGL(l2_step_kernel<3>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, grad, step_size, out, ws, B, E, eps, a, oma);
  hipLaunchKernelGGL(l2_step_kernel<4>, grid, dim3(kBlock), 0, s, x, x_adv, x_adv_old, grad, step_size, out, ws, B, E, eps, a, oma);
  return launch_status();
}

int apgd_loss_pred(const void* logits, int dtype, int64_t ld, const int64_t* y_hard, const float* y_soft,
                   int loss_kind, float* loss, uint8_t* pred, void* dlogits, int64_t B, int64_t n_cls, void* stream) {
  if (B < 0 || n_cls <= 0 || ld < n_cls) return APGD_ERR_SIZE;
  if (B == 0) return APGD_OK;
  if (!logits || !loss || !pred) return APGD_ERR_NULL;
  if ((y_hard == nullptr) == (y_soft == nullptr)) return APGD_ERR_ARG;
  if (loss_kind != 0 && loss_kind != 1) return APGD_ERR_ARG;
  if (loss_kind == 1 && (!y_hard || n_cls < 3)) return APGD_ERR_ARG;     // dlr: hard labels, >= 3 classes (:99-104)
  if (n_cls > 0x7ffffffe) return APGD_ERR_SIZE;
  hipStream_t s = as_stream(stream);
  const dim3 grid(static_cast<unsigned>((B + kBlock / kWave - 1) / (kBlock / kWave)));
  if (loss_kind == 1) {
    switch (dtype) {
      case APGD_F32:
        hipLaunchKernelGGL(dlr_pred_kernel<float>, grid, dim3(kBlock), 0, s, static_cast<const float*>(logits), ld, y_hard,
                           loss, pred, static_cast<float*>(dlogits), B, n_cls);
        break;
      case APGD_BF16:
        hipLaunchKernelGGL(dlr_pred_kernel<uint16_t>, grid, dim3(kBlock), 0, s, static_cast<const uint16_t*>(logits), ld,
                           y_hard, loss, pred, static_cast<uint16_t*>(dlogits), B, n_cls);
        break;
      case APGD_F16:
        hipLaunchKernelGGL(dlr_pred_kernel<_Float16>, grid, dim3(kBlock), 0, s, static_cast<const _Float16*>(logits), ld,
                           y_hard, loss, pred, static_cast<_Float16*>(dlogits), B, n_cls);
        break;
      default: return APGD_ERR_DTYPE;
    }
    return launch_status();
  }
  switch (dtype) {
    case APGD_F32:
      hipLaunchKernelGGL(ce_pred_kernel<float>, grid, dim3(kBlock), 0, s, static_cast<const float*>(logits), ld, y_hard,
                         y_soft, loss, pred, static_cast<float*>(dlogits), B, n_cls);
      break;
    case APGD_BF16:
      hipLaunchKernelGGL(ce_pred_kernel<uint16_t>, grid, dim3(kBlock), 0, s, static_cast<const uint16_t*>(logits), ld,
                         y_hard, y_soft, loss, pred, static_cast<uint16_t*>(dlogits), B, n_cls);
      break;
    case APGD_F16:
      hipLaunchKernelGGL(ce_pred_kernel<_Float16>, grid, dim3(kBlock), 0, s, static_cast<const _Float16*>(logits), ld,
                         y_hard, y_soft, loss, pred, static_cast<_Float16*>(dlogits), B, n_cls);
      break;
    default: return APGD_ERR_DTYPE;
  }
  return launch_status();
}

int apgd_loss_pred_targeted(const void* logits, int dtype, int64_t ld, const int64_t* y_hard, const int64_t* y_target,
                            float* loss, uint8_t* pred, void* dlogits, int64_t B, int64_t n_cls, void* stream) {
  if (B < 0 || n_cls <= 0 || ld < n_cls) return APGD_ERR_SIZE;
  if (B == 0) return APGD_OK;
  if (!logits || !loss || !pred || !y_hard || !y_target) return APGD_ERR_NULL;
  if (n_cls < 4) return APGD_ERR_ARG;                                     // :110 indexes the 4th largest logit
  if (n_cls > 0x7ffffffe) return APGD_ERR_SIZE;
  hipStream_t s = as_stream(stream);
  const dim3 grid(static_cast<unsigned>((B + kBlock / kWave - 1) / (kBlock / kWave)));
  switch (dtype) {
    case APGD_F32:
      hipLaunchKernelGGL(dlr_targeted_pred_kernel<float>, grid, dim3(kBlock), 0, s, static_cast<const float*>(logits), ld,
                         y_hard, y_target, loss, pred, static_cast<float*>(dlogits), B, n_cls);
      break;
    case APGD_BF16:
      hipLaunchKernelGGL(dlr_targeted_pred_kernel<uint16_t>, grid, dim3(kBlock), 0, s, static_cast<const uint16_t*>(logits),
                         ld, y_hard, y_target, loss, pred, static_cast<uint16_t*>(dlogits), B, n_cls);
      break;
    case APGD_F16:
      hipLaunchKernelGGL(dlr_targeted_pred_kernel<_Float16>, grid, dim3(kBlock), 0, s, static_cast<const _Float16*>(logits),
                         ld, y_hard, y_target, loss, pred, static_cast<_Float16*>(dlogits), B, n_cls);
      break;
    default: return APGD_ERR_DTYPE;
  }
  return launch_status();
}

int apgd_state_update(const float* loss, const uint8_t* pred, uint8_t* acc, float* loss_best, float* loss_best_last,
                      float* reduced_last, float* step_size, float* loss_steps, uint8_t* flags, int64_t B, int32_t K,
                      int32_t i, int32_t do_check, int32_t k, float thr, void* stream) {
  if (B < 0 || K <= 0 || i < 0 || i >= K) return APGD_ERR_SIZE;
  if (do_check && (k <= 0 || k > i + 1)) return APGD_ERR_ARG;
  if (B == 0) return APGD_OK;
  if (!loss || !pred || !acc || !loss_best || !loss_best_last || !reduced_last || !step_size || !loss_steps || !flags)
    return APGD_ERR_NULL;
  const dim3 grid(static_cast<unsigned>((B + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(state_update_kernel, grid, dim3(kBlock), 0, as_stream(stream), loss, pred, acc, loss_best,
                     loss_best_last, reduced_last, step_size, loss_steps, flags, B, K, i, do_check, k, thr);
  return launch_status();
}

int apgd_track_rows(const uint8_t* flags, float* x_adv, void* grad, float* x_best, void* grad_best, float* x_best_adv,
                    int32_t grad_elt, int64_t B, int64_t E, int32_t final_iter, void* stream) {
  if (B < 0 || E < 0) return APGD_ERR_SIZE;
  if (B == 0 || E == 0) return APGD_OK;
  if (!flags || !x_adv || !x_best || !x_best_adv) return APGD_ERR_NULL;
  if ((grad == nullptr) != (grad_best == nullptr)) return APGD_ERR_ARG;
  if (grad && grad_elt != 4 && grad_elt != 2 && grad_elt != 1) return APGD_ERR_DTYPE;
  if (B > 65535) return APGD_ERR_SIZE;
  const int ge = grad ? grad_elt : 4;
  const bool vec16 = ((E * 4) % 16 == 0) && ((E * ge) % 16 == 0) && aligned16(x_adv) && aligned16(x_best) &&
                     aligned16(x_best_adv) && aligned16(grad) && aligned16(grad_best);
  const int64_t want = (256 * 8 * 2 + B - 1) / B;
  const dim3 grid(blocks_for(vec16 ? E / 4 : E * 2, static_cast<int64_t>(kBlock) * 2, want < 1 ? 1 : want),
                  static_cast<unsigned>(B));
  hipLaunchKernelGGL(track_rows_kernel, grid, dim3(kBlock), 0, as_stream(stream), flags, x_adv,
                     static_cast<uint8_t*>(grad), x_best, static_cast<uint8_t*>(grad_best), x_best_adv, ge, E,
                     final_iter, vec16 ? 1 : 0);
  return launch_status();
}

int apgd_check_imgs_f32(const float* adv, const float* x, float* out, int64_t B, int64_t E, void* stream) {
  if (B < 0 || E < 0) return APGD_ERR_SIZE;
  if (B == 0) return APGD_OK;
  if (!adv || !x || !out) return APGD_ERR_NULL;
  hipLaunchKernelGGL(check_imgs_kernel, dim3(static_cast<unsigned>(B)), dim3(kBlock), 0, as_stream(stream), adv, x, out, E);
  return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------- adv.attack = fgsm (fgsm_train.py:72-100, main.py:836-842)
// The other value of the trainer's attack selector: random start, ONE signed gradient step, projection.  Two element-wise
// passes around the single forward / backward; the gradient may arrive as fp32, bf16 or the stem kernel's int8 signs.
namespace {
__device__ __forceinline__ float sign3(float g) { return static_cast<float>(g > 0.0f) - static_cast<float>(g < 0.0f); }

// x_adv = x + ((2 t - 1) * eps) * noise   (:81-82; every operation rounded on its own, as the reference's tensor expression)
__global__ __launch_bounds__(kBlock) void fgsm_start_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            float* __restrict__ out, int64_t n, float eps, float noise, int clamp) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < n; e += stride) {
    float v = 2.0f * t[e];
    v = v - 1.0f;
    v = v * eps;
    v = v * noise;
    v = x[e] + v;
    out[e] = clamp ? clamp01(v) : v;
  }
}

// out = x_adv + (alpha eps) sign(g);  unless skip_projection: out = clamp01(x + clamp(out - x, -eps, eps))   (:95-98)
template <typename GT>
__global__ __launch_bounds__(kBlock) void fgsm_step_kernel(const float* __restrict__ x, const float* __restrict__ xa,
                                                           const GT* __restrict__ g, float* __restrict__ out, int64_t n,
                                                           float alpha_eps, float eps, int project) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < n; e += stride) {
    float v = alpha_eps * sign3(Elt<GT>::load(g, e));
    v = xa[e] + v;
    if (project) {
      const float xc = x[e];
      float d = v - xc;
      d = fminf(fmaxf(d, -eps), eps);
      v = clamp01(xc + d);
    }
    out[e] = v;
  }
}
}  // namespace

extern "C" {

int apgd_fgsm_start_f32(const float* x, const float* t, float* x_adv, int64_t n, float eps, float noise_level, int32_t clamp,
                        void* stream) {
  if (n < 0) return APGD_ERR_SIZE;
  if (n == 0) return APGD_OK;
  if (!x || !t || !x_adv) return APGD_ERR_NULL;
  hipLaunchKernelGGL(fgsm_start_kernel, dim3(blocks_for(n, kBlock, 16384)), dim3(kBlock), 0, as_stream(stream), x, t, x_adv, n, eps,
                     noise_level, clamp);
  return launch_status();
}

int apgd_fgsm_step_f32(const float* x, const float* x_adv, const void* grad, int32_t grad_dtype, float* out, int64_t n,
                       float alpha_eps, float eps, int32_t project, void* stream) {
  if (n < 0) return APGD_ERR_SIZE;
  if (n == 0) return APGD_OK;
  if (!x || !x_adv || !grad || !out) return APGD_ERR_NULL;
  const dim3 grid(blocks_for(n, kBlock, 16384)), block(kBlock);
  hipStream_t s = as_stream(stream);
  switch (grad_dtype) {
    case APGD_F32:
      hipLaunchKernelGGL(fgsm_step_kernel<float>, grid, block, 0, s, x, x_adv, static_cast<const float*>(grad), out, n, alpha_eps, eps, project);
      break;
    case APGD_BF16:
      hipLaunchKernelGGL(fgsm_step_kernel<uint16_t>, grid, block, 0, s, x, x_adv, static_cast<const uint16_t*>(grad), out, n, alpha_eps, eps,
                         project);
      break;
    case APGD_I8:
      hipLaunchKernelGGL(fgsm_step_kernel<int8_t>, grid, block, 0, s, x, x_adv, static_cast<const int8_t*>(grad), out, n, alpha_eps, eps,
                         project);
      break;
    default: return APGD_ERR_DTYPE;
  }
  return launch_status();
}

}  // extern "C"
