// lds_probe.cpp - LDS read bandwidth of one gfx950 CU for the access patterns of the fused MLP kernels.
//   ds_read_b128, lane l reads 16 bytes at base + 16 l (one contiguous KiB per instruction: the MFMA operand fragments),
//   every wave its own 4 KiB / all waves the same 4 KiB (the kernels' case) / lanes permuted inside the KiB; ds_read_b64, ds_read_b32.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o lds_probe lds_probe.cpp ; run: lds_probe [assumed GHz]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int PAT>
__global__ __launch_bounds__(1024) void lds_read(unsigned* out, unsigned long long* cyc, int iters) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const unsigned base = (PAT == 4 ? 0 : (wave & 15) * 4096);
  const unsigned addr = base + (PAT == 1 ? 8 * lane : PAT == 2 ? 4 * lane : PAT == 3 ? 16 * (lane ^ 5) : 16 * lane);
  u32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0 || PAT == 3 || PAT == 4) {
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));
    } else if (PAT == 1) {
      u32x2 b0, b1, b2, b3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
      a0.x += b0.x + b1.x + b2.x + b3.x;
    } else {
      unsigned b0, b1, b2, b3;
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
      a0.x += b0 + b1 + b2 + b3;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.z + a3.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT>
void run(const char* what, int bytes_per_inst, unsigned* out, unsigned long long* cyc) {
  printf("%-50s", what);
  for (int waves : {1, 4, 8, 12, 16}) {
    const int iters = 4000;
    hipLaunchKernelGGL((lds_read<PAT>), dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
    hipLaunchKernelGGL((lds_read<PAT>), dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += static_cast<double>(h[i]);
    c /= 256;                                                     // shader cycles wave 0 of a CU spent in the loop
    const double bytes = static_cast<double>(iters) * 4 * waves * bytes_per_inst;
    printf("  %2d waves: %6.1f B/clk (%5.0f clk per 4 reads)", waves, bytes / c, c / iters);
  }
  printf("\n");
}

int main() {
  unsigned* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(unsigned));
  hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("LDS read bandwidth per CU in bytes per shader clock (s_memtime), 4 reads in flight per wave, then s_waitcnt\n");
  run<0>("ds_read_b128 linear, each wave its own 4 KiB", 1024, out, cyc);
  run<4>("ds_read_b128 linear, all waves the same 4 KiB", 1024, out, cyc);
  run<3>("ds_read_b128 permuted lanes", 1024, out, cyc);
  run<1>("ds_read_b64 linear", 512, out, cyc);
  run<2>("ds_read_b32 linear", 256, out, cyc);
  return 0;
}
