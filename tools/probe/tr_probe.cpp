// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which 16-bit element of the LDS image each lane receives.
// LDS holds u16 element e at index e (value = index); every lane hands the instruction the byte address pattern[lane]; the
// four u16 results of every lane are printed.  Patterns:
//   0: addr = lane * 8                 (64 lanes read 512 contiguous bytes)
//   1: addr = (lane % 16) * 8 + (lane / 16) * 1024
//   2: 16-lane group g reads rows 4g..4g+3 of a [16][64]-element image (row stride 128 B), lane i: row i / 4, column chunk i % 4
//   3: the same with lane i: row i % 4, column chunk i / 4
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.cpp -o /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void probe(const uint32_t* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<uint16_t>(i);
  __syncthreads();
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)&lds[0]));
  uint32_t a = base + addr[threadIdx.x];
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}

int main() {
  uint32_t* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
  for (int pat = 0; pat < 4; ++pat) {
    std::vector<uint32_t> a(64);
    for (int l = 0; l < 64; ++l) {
      const int g = l / 16, i = l % 16;
      if (pat == 0) a[l] = l * 8;
      else if (pat == 1) a[l] = i * 8 + g * 1024;
      else if (pat == 2) a[l] = ((4 * g + i / 4) * 64 + (i % 4) * 4) * 2;
      else a[l] = ((4 * g + i % 4) * 64 + (i / 4) * 4) * 2;
    }
    hipMemcpy(d_addr, a.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<uint16_t> o(256);
    hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (element indices; lane: addr/2 -> 4 results)\n", pat);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: addr_elem %5u -> %5u %5u %5u %5u\n", l, a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  return 0;
}
