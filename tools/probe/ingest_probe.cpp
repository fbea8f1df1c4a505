// ingest_probe.cpp - what does a CU take in per clock, and from where?  (round 6; `hipcc --offload-arch=gfx950 -O3 ingest_probe.cpp`)
//
// Every GEMM-shaped kernel of this library - and hipBLASLt's - stops at ~7.5 TB/s of staged operand bytes over the chip
// (13 - 15 B / clock / CU).  This probe streams 1 KiB pieces (64 lanes x 16 B, the shape of those kernels' loads) from a region of a
// chosen size - 1 MiB per XCD (L2 hits), 64 MiB (Infinity Cache), 4 GiB (HBM) - into a CU in three ways:
//   mode 0  LDS-DMA          buffer_load_dwordx4 ... lds   (what the kernels do)
//   mode 1  to registers     global_load_dwordx4, values folded into one register
//   mode 2  registers + LDS  global_load_dwordx4 then ds_write_b128
// with `depth` pieces in flight per wavefront and `waves` wavefronts per workgroup, one workgroup per CU (grid 256) or two.
// share = 1: all workgroups read the SAME addresses in step (the pattern of workgroups that share an operand tile); 0: every
// workgroup walks its own slice of the region.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ void dma_lds16(uint32_t lds_dst, uint32_t voff, u32x4 rs, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void ingest_kernel(const uint8_t* __restrict__ src, uint64_t region, uint32_t pieces_per_wave, int share,
                                                    uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  // the workgroup's walk: piece p of wave w sits at ((base + (p * nw + w)) * 1024) % region
  const uint64_t wg_pieces = static_cast<uint64_t>(pieces_per_wave) * nw;
  const uint64_t base = share ? 0 : static_cast<uint64_t>(blockIdx.x) * wg_pieces;
  const uint64_t region_pieces = region >> 10;                           // a power of two
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)lds));
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    // 32-bit offsets: windows of 2 GiB re-based per piece through the scalar offset are not needed - the base pointer moves instead
    for (uint32_t p = 0; p < pieces_per_wave; p += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const uint64_t piece = (base + static_cast<uint64_t>(p + d) * nw + wave) & (region_pieces - 1);
        const uint64_t addr = a + (piece << 10);
        const u32x4 rs = {static_cast<uint32_t>(addr), static_cast<uint32_t>(addr >> 32) & 0xffffu, 1024u, 0x00020000u};
        dma_lds16(lds0 + static_cast<uint32_t>((wave * DEPTH + d) * 1024), static_cast<uint32_t>(lane * 16), rs, 0u);
      }
      wait_vmcnt<0>();
    }
  } else {
    for (uint32_t p = 0; p < pieces_per_wave; p += DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const uint64_t piece = (base + static_cast<uint64_t>(p + d) * nw + wave) & (region_pieces - 1);
        v[d] = *reinterpret_cast<const u32x4*>(src + (piece << 10) + lane * 16);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 1) acc ^= v[d];
        else *reinterpret_cast<u32x4*>(lds + (wave * DEPTH + d) * 1024 + lane * 16) = v[d];
      }
    }
  }
  if (MODE != 1) {
    __syncthreads();
    acc = *reinterpret_cast<const u32x4*>(lds + ((threadIdx.x * 16) % (nw * DEPTH * 1024)));
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;       // keeps the loads alive
}

template <int MODE, int DEPTH>
double run(const uint8_t* src, uint64_t region, int grid, int waves, uint32_t pieces_per_wave, int share, uint32_t* sink) {
  const size_t lds = static_cast<size_t>(waves) * DEPTH * 1024;
  auto k = ingest_kernel<MODE, DEPTH>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, 0, src, region, pieces_per_wave, share, sink);
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < 5; ++i) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, 0, src, region, pieces_per_wave, share, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main(int argc, char** argv) {
  const uint64_t GiB = 1ull << 30;
  uint8_t* buf; uint32_t* sink;
  if (hipMalloc(&buf, 4 * GiB) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, 4 * GiB);
  const uint32_t ppw = 4096;                                          // pieces per wave: 4 MiB per wave
  printf("%-28s %-8s %5s %5s %5s %6s %10s %9s %12s\n", "region", "mode", "grid", "waves", "depth", "share", "ms", "TB/s", "B/clk/CU@2.1");
  struct R { const char* name; uint64_t bytes; } regions[] = {{"2 MiB (in every XCD's L2)", 2ull << 20}, {"64 MiB (Infinity Cache)", 64ull << 20}, {"4 GiB (HBM)", 4 * GiB}};
  const char* mn[] = {"lds-dma", "regs", "regs+lds"};
  for (auto& r : regions)
    for (int share = 0; share < 2; ++share)
      for (int grid : {256, 512})
        for (int waves : {4, 8}) {
          if (grid == 512 && waves == 8 && false) continue;
#define ONE(MODE, DEPTH)                                                                                                              \
  {                                                                                                                                   \
    const double ms = run<MODE, DEPTH>(buf, r.bytes, grid, waves, ppw, share, sink);                                                  \
    const double bytes = static_cast<double>(grid) * waves * ppw * 1024.0;                                                            \
    printf("%-28s %-8s %5d %5d %5d %6d %10.3f %9.2f %12.1f\n", r.name, mn[MODE], grid, waves, DEPTH, share, ms, bytes / ms / 1e9,    \
           bytes / (ms * 1e-3) / 256.0 / 2.1e9);                                                                                      \
  }
          ONE(0, 4) ONE(0, 9) ONE(0, 16) ONE(1, 4) ONE(1, 9) ONE(1, 16) ONE(2, 9)
#undef ONE
        }
  return 0;
}
