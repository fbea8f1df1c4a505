// epi_probe.cpp - how fast does a wavefront move its 32-row x C fp32 output tile (read the residual, write the result) with
//   (a) "row-contiguous": 16 B per lane, consecutive lanes on consecutive 16-byte chunks of a row (what the fused LN+MLP kernels do
//       today, after an LDS transpose of the accumulators: 1 KiB contiguous per instruction), and
//   (b) "accumulator order of the swapped GEMM2": lane = row (l32), the two half-waves on adjacent 16-byte quads - 32 B per row and
//       instruction, 32 rows per instruction (what O^T = W2 . H^T would leave in the registers: no LDS transpose needed).
// A design question for the LDS-resident-weights kernel of DESIGN.md section 8: is pattern (b) good enough to drop the transpose?
// Build: hipcc --offload-arch=gfx950 -O3 -w -o epi_probe epi_probe.cpp ; run: epi_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int C, int PAT>
__global__ __launch_bounds__(256) void epi(const float* __restrict__ x, float* __restrict__ out, long M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
  const long m0 = (static_cast<long>(blockIdx.x) * 4 + wave) * 32;
  if (m0 >= M) return;
  if (PAT == 0) {
    // 32 rows x C floats = 8 C chunks of 16 B, 64 per instruction
#pragma unroll
    for (int i = 0; i < C / 8; ++i) {
      const int idx = i * 64 + lane, rr = idx / (C / 4), c4 = (idx - rr * (C / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4*>(x + (m0 + rr) * C + c4);
      *reinterpret_cast<float4*>(out + (m0 + rr) * C + c4) = make_float4(v.x + 1.f, v.y + 1.f, v.z + 1.f, v.w + 1.f);
    }
  } else {
    // per 32-channel block cb and register quad g: channels cb*32 + 8 g + 4 half .. +3 of row l32
#pragma unroll
    for (int cb = 0; cb < C / 32; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c4 = cb * 32 + 8 * g + 4 * half;
        const float4 v = *reinterpret_cast<const float4*>(x + (m0 + l32) * C + c4);
        *reinterpret_cast<float4*>(out + (m0 + l32) * C + c4) = make_float4(v.x + 1.f, v.y + 1.f, v.z + 1.f, v.w + 1.f);
      }
  }
}

template <int C, int PAT>
void run(const char* what, const float* x, float* out, long M) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(static_cast<unsigned>((M + 127) / 128));
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((epi<C, PAT>), grid, dim3(256), 0, 0, x, out, M);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const double bytes = 2.0 * M * C * 4;
  printf("C=%3d M=%7ld %-44s avg %7.1f us  %6.0f GB/s   best %7.1f us\n", C, M, what, sum / 10 * 1e3, bytes / (sum / 10 * 1e-3) / 1e9, best * 1e3);
}

int main() {
  const long M96 = 256L * 56 * 56, M192 = 256L * 28 * 28, M384 = 256L * 14 * 14;
  float *x, *out;
  hipMalloc(&x, M96 * 96 * 4); hipMalloc(&out, M96 * 96 * 4);
  hipMemset(x, 0, M96 * 96 * 4);
  run<96, 0>("row-contiguous 16 B chunks (today)", x, out, M96);
  run<96, 1>("lane = row, 32 B per row and instruction", x, out, M96);
  run<192, 0>("row-contiguous 16 B chunks (today)", x, out, M192);
  run<192, 1>("lane = row, 32 B per row and instruction", x, out, M192);
  run<384, 0>("row-contiguous 16 B chunks (today)", x, out, M384);
  run<384, 1>("lane = row, 32 B per row and instruction", x, out, M384);
  return 0;
}
