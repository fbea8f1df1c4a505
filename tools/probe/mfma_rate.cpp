// MFMA issue-rate calibration on the box: (a) register operands only, (b) one ds_read_b128 operand per MFMA,
// (c) as (b) with 2 accumulators.  hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.cpp -o tools/probe/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 b; for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.01f * (lane + e));
  bf16x8 areg = b;
  const unsigned char* sl = lds + lane * 16;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(sl));            // opaque per iteration: the LDS reads cannot be hoisted out of the loop
#pragma unroll
    for (int i = 0; i < 48; ++i) {
      bf16x8 a;
      if (MODE == 0) a = areg; else a = *reinterpret_cast<const bf16x8*>(sl + i * 1024);
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
    }
  }
  float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int wgs) {
  float* out; hipMalloc(&out, wgs * 256 * 4);
  const int iters = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NACC><<<wgs, 256, 48 * 1024>>>(out, 10);
  hipEventRecord(e0);
  k<MODE, NACC><<<wgs, 256, 48 * 1024>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * 48.0 * iters * wgs * 4;
  printf("%-40s wgs %5d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz if 1 wave/SIMD)\n", name, wgs, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (48.0 * iters * (wgs / 256.0)));
  hipFree(out);
}

int main() {
  run<0, 1>("reg operands, 1 acc chain", 256);
  run<0, 4>("reg operands, 4 acc chains", 256);
  run<1, 1>("ds_read_b128 per MFMA, 1 chain", 256);
  run<1, 2>("ds_read_b128 per MFMA, 2 chains", 256);
  run<1, 4>("ds_read_b128 per MFMA, 4 chains", 256);
  run<1, 4>("ds_read_b128 per MFMA, 4 chains, 2 WG/CU", 512);
  return 0;
}
