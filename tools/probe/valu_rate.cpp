// VALU issue rates on gfx950: wave-instructions per SIMD and nanosecond for v_dot2c_f32_bf16, v_fma_f32, v_pk_fma_f32 and a mix,
// at 1 / 2 / 3 wavefronts per SIMD (one workgroup of 256 x waves-per-SIMD threads per CU).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int MODE>
__global__ __launch_bounds__(768) void rate_kernel(float* out, int iters, uint32_t seed) {
  float acc[8];
  uint32_t a[4], b[4];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9f;
  for (int i = 0; i < 4; ++i) { a[i] = seed + i * 0x10001u; b[i] = seed * 3u + i; }
  f32x2_t pa[4], pb = {1.0001f, 0.9999f};
  for (int i = 0; i < 4; ++i) pa[i] = f32x2_t{acc[i], acc[i + 4]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[i & 3]), __builtin_bit_cast(bf16x2_t, b[r & 3]), acc[i], false);
        else if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[r & 3]));
        else if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i & 3]) : "v"(pb), "v"(pa[(i + 1) & 3]));
        else if constexpr (MODE == 3) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[r & 3]));
        else if constexpr (MODE == 4) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i & 3]) : "v"(b[r & 3]));
        else if constexpr (MODE == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i & 3]) : "v"(acc[i]), "v"(acc[(i + 1) & 7]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i];
  for (int i = 0; i < 4; ++i) s += pa[i].x + pa[i].y + __uint_as_float(a[i] & 0x3fffffffu);
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out) {
  for (int wps = 1; wps <= 3; ++wps) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 64.0 * iters * wps;                 // per SIMD
    printf("%-18s waves/SIMD %d: %.3f ms, %.3f wave-instructions per ns per SIMD (%.2f cycles each at 2.4 GHz)\n", name, wps, ms, insts / (ms * 1e6),
           ms * 1e6 * 2.4 / insts);
  }
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<0>("dot2 (builtin)", out);
  run<3>("v_dot2c_f32_bf16", out);
  run<1>("v_fma_f32", out);
  run<2>("v_pk_fma_f32", out);
  run<4>("v_and_b32", out);
  run<5>("v_cvt_pk_bf16_f32", out);
  return 0;
}
