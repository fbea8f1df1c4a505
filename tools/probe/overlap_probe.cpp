// overlap_probe.cpp - do the matrix pipe and the VALU of one SIMD overlap on gfx950, and how?
// Each wavefront runs ITER iterations of a body with NM independent-chain MFMAs (32x32x16 bf16) and / or NV packed-fp32 FMAs
// in NC independent chains; 256 workgroups (one per CU) of 4*W wavefronts (W per SIMD).
//   mode 0: MFMA only          mode 1: VALU only          mode 2: both in the SAME wavefront, interleaved in the source
//   mode 3: even wavefronts MFMA only, odd wavefronts VALU only (needs W >= 2: waves w and w+4 share a SIMD)
// Output: cycles per iteration per SIMD (from s_memtime) for each (mode, W).
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE, int NM, int NV>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f32x2 v[8];
  for (int k = 0; k < 8; ++k) v[k] = (f32x2){0.5f + lane * 1e-3f + k, 0.25f};
  const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) { const int k = (i * (NV / NM) + j) & 7; v[k] = __builtin_elementwise_fma(v[k], m, c); }
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_elementwise_fma(v[j & 7], m, c);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][lane & 15];
  for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE, int NM, int NV>
void run(int W, float* out, unsigned long long* cyc, const char* what) {
  const int iters = 2000, waves = 4 * W;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NM, NV>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NM, NV>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * waves);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto x : h) avg += static_cast<double>(x);
  avg /= h.size();
  // s_memtime / readcyclecounter ticks at a constant 100 MHz on some parts: report wall time per iteration per SIMD as well
  printf("%-34s W=%d  NM=%2d NV=%3d  wall %8.1f us  -> %7.1f ns per iteration per wave-slot   (counter: %.0f ticks/iter)\n", what, W, NM, NV,
         ms * 1e3, ms * 1e6 / iters, avg / iters);
}


// ---- exact streams (inline asm, nothing for the compiler to split or move): 1 MFMA followed by NV packed FMAs, 12 times per iteration
// ACC: 0 = no MFMA, 1 = accumulators in ArchVGPRs, 2 = accumulators in AccVGPRs.  SPLIT: waves whose slot on their SIMD is odd run
// only the VALU part and the others only the MFMA part (slot = order of arrival on the SIMD, from HW_ID).
__device__ inline unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }

template <int NV, int ACC, bool SPLIT>
__global__ __launch_bounds__(1024) void exact(float* out, unsigned* ids, int iters) {
  __shared__ unsigned simd_of[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f32x2 v[8];
  for (int k = 0; k < 8; ++k) v[k] = (f32x2){0.5f + lane * 1e-3f + k, 0.25f};
  f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  asm volatile("" : "+v"(m), "+v"(c));
  const unsigned id = hw_id();
  if (lane == 0) simd_of[wave] = (id >> 4) & 3;
  __syncthreads();
  int slot = 0;
  for (int w = 0; w < wave; ++w) slot += simd_of[w] == simd_of[wave];
  const bool do_m = ACC != 0 && (!SPLIT || (slot & 1) == 0), do_v = !SPLIT || (slot & 1) == 1;
  if (lane == 0 && blockIdx.x == 0) ids[wave] = id;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (do_m) {
        if (ACC == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
        if (ACC == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
      }
      if (do_v) {
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * NV + j) & 7]) : "v"(m), "v"(c));
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][lane & 15];
  for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_ns_per_cycle = 0.0;

template <int NV, int ACC, bool SPLIT>
double run_exact(int W, float* out, unsigned* ids) {      // ns per group per SIMD
  const int iters = 2000, waves = 4 * W;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((exact<NV, ACC, SPLIT>), dim3(256), dim3(64 * waves), 0, 0, out, ids, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((exact<NV, ACC, SPLIT>), dim3(256), dim3(64 * waves), 0, 0, out, ids, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e6 / iters / 12.0;
}

template <int NV>
void row(float* out, unsigned* ids) {
  printf("NV=%2d |", NV);
  for (int W = 1; W <= 3; ++W) {
    const double c = g_ns_per_cycle;
    printf("  W=%d: valu %6.1f  +mfma(v) %6.1f  +mfma(a) %6.1f", W, NV ? run_exact<NV, 0, false>(W, out, ids) / c : 0.0,
           run_exact<NV, 1, false>(W, out, ids) / c, run_exact<NV, 2, false>(W, out, ids) / c);
    if (W >= 2) printf("  split(v) %6.1f  split(a) %6.1f", run_exact<NV, 1, true>(W, out, ids) / c, run_exact<NV, 2, true>(W, out, ids) / c);
    printf(" |");
  }
  printf("\n");
}

void exact_table(float* out, unsigned* ids) {
  g_ns_per_cycle = run_exact<0, 1, false>(1, out, ids) / 32.0;   // one wave per SIMD issuing MFMAs back to back: 32 cycles each
  printf("\ncalibration: %.4f ns per cycle (%.2f GHz) from back-to-back MFMAs\n", g_ns_per_cycle, 1.0 / g_ns_per_cycle);
  hipLaunchKernelGGL((exact<1, 1, false>), dim3(256), dim3(64 * 12), 0, 0, out, ids, 1);
  hipDeviceSynchronize();
  unsigned h[16];
  hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
  printf("workgroup 0, 12 waves: SIMD of wave w =");
  for (int w = 0; w < 12; ++w) printf(" %u", (h[w] >> 4) & 3);
  printf("\n\nCYCLES per group (1 MFMA 32x32x16 + NV v_pk_fma_f32) per SIMD WALL time, W waves per SIMD; for W waves all running the same\n"
         "stream the figure is for W groups.  split: every other wave of a SIMD runs only the MFMAs, the others only the VALU part.\n");
  if (getenv("OVERLAP_FULL")) {
    row<0>(out, ids); row<1>(out, ids); row<2>(out, ids); row<3>(out, ids); row<4>(out, ids); row<5>(out, ids); row<6>(out, ids);
    row<7>(out, ids); row<8>(out, ids); row<10>(out, ids); row<12>(out, ids); row<16>(out, ids);
  }
}


// ---- phase-structured streams, the shape of a fused MLP wave: NM MFMAs back to back, then NV packed FMAs, repeated; W waves per
// SIMD; wave slot s of a SIMD starts s * stagger_cycles late (0 = all in phase).
template <int NM, int NV>
__global__ __launch_bounds__(1024) void phases(float* out, int iters, int stagger) {
  __shared__ unsigned simd_of[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f32x2 v[8];
  for (int k = 0; k < 8; ++k) v[k] = (f32x2){0.5f + lane * 1e-3f + k, 0.25f};
  f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  asm volatile("" : "+v"(m), "+v"(c));
  if (lane == 0) simd_of[wave] = (hw_id() >> 4) & 3;
  __syncthreads();
  int slot = 0;
  for (int w = 0; w < wave; ++w) slot += simd_of[w] == simd_of[wave];
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)(slot * stagger)) __builtin_amdgcn_s_sleep(1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "v"(c));
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][lane & 15];
  for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV>
void run_phases(float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("\nphases: %d MFMAs then %d v_pk_fma_f32 per iteration and wave; additive law = W * (%d*32 + %d*3.8) = W * %.0f cycles\n", NM, NV, NM, NV,
         NM * 32 + NV * 3.8);
  for (int W = 1; W <= 4; ++W)
    for (int stagger : {0, 150, 300, 450}) {
      if (W == 1 && stagger) continue;
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((phases<NM, NV>), dim3(256), dim3(256 * W), 0, 0, out, iters, stagger);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("  W=%d stagger %3d cycles: %7.0f cycles per iteration per SIMD = %6.0f per wave\n", W, stagger, best * 1e6 / iters / g_ns_per_cycle,
             best * 1e6 / iters / g_ns_per_cycle / W);
    }
}


// ---- which VALU instructions hide behind an MFMA?  1 MFMA + NV instructions of kind OP per group, one wave per SIMD and three.
// OP: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_exp_f32, 3 v_and_b32, 4 v_cvt_pk_bf16_f32, 5 v_pk_mul_f32, 6 v_add_u32, 7 v_pk_fma_f16
template <int NV, int OP, bool WITH_MFMA>
__global__ __launch_bounds__(1024) void kinds(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f32x2 v[8];
  for (int k = 0; k < 8; ++k) v[k] = (f32x2){0.5f + lane * 1e-3f + k, 0.25f};
  f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
  asm volatile("" : "+v"(m), "+v"(c));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (WITH_MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        f32x2& r = v[(i * NV + j) & 7];
        if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x) : "v"(m.x), "v"(c.x));
        if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r.x));
        if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r.x) : "v"(m.x));
        if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r.x) : "v"(m.x));
        if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(m));
        if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.x) : "v"(m.x));
        if (OP == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r.x) : "v"(m.x), "v"(c.x));
        if (OP == 8) asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(v[0].x) : "v"(m.x), "s"(1e-3f));            // ONE dependent chain, VOP3 modifiers + SGPR
        if (OP == 9) asm volatile("v_fma_f32 %0, %0, |%1|, %2\n s_nop 0" : "+v"(v[0].x) : "v"(m.x), "s"(1e-3f));   // ... with the s_nop hipcc puts after inline asm
        if (OP == 10) asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(v[j & 1].x) : "v"(m.x), "s"(1e-3f));        // two interleaved chains
        if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0].x) : "v"(m.x), "v"(c.x));                // one dependent chain, plain operands
        if (OP == 12) asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(v[j & 3].x) : "v"(m.x), "s"(1e-3f));        // four interleaved chains
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][lane & 15];
  for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int OP, bool WITH_MFMA>
double time_kinds(int W, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kinds<NV, OP, WITH_MFMA>), dim3(256), dim3(256 * W), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e6 / iters / 12.0 / g_ns_per_cycle / W;
}

template <int OP>
void kind_row(const char* name, float* out) {
  printf("%-20s", name);
  for (int W : {1, 3})
    printf(" | W=%d  NV=4: alone %5.1f with MFMA %5.1f   NV=8: alone %5.1f with MFMA %5.1f   NV=16: alone %5.1f with MFMA %5.1f", W,
           time_kinds<4, OP, false>(W, out), time_kinds<4, OP, true>(W, out), time_kinds<8, OP, false>(W, out), time_kinds<8, OP, true>(W, out),
           time_kinds<16, OP, false>(W, out), time_kinds<16, OP, true>(W, out));
  printf("\n");
}

void kinds_table(float* out) {
  printf("\ncycles per group (1 MFMA + NV instructions of one kind) per wave; 'alone' = without the MFMA; MFMA alone = 32\n");
  kind_row<0>("v_pk_fma_f32", out); kind_row<1>("v_fma_f32", out); kind_row<2>("v_exp_f32", out); kind_row<3>("v_and_b32", out);
  kind_row<8>("fma chain |abs| sgpr", out); kind_row<9>("fma chain + s_nop 0", out); kind_row<10>("2 fma chains", out);
  kind_row<12>("4 fma chains", out); kind_row<11>("fma chain plain", out);
  kind_row<4>("v_cvt_pk_bf16_f32", out); kind_row<5>("v_pk_mul_f32", out); kind_row<6>("v_add_u32", out); kind_row<7>("v_pk_fma_f16", out);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  for (int W = 1; W <= 3; ++W) {
    run<0, 12, 112>(W, out, cyc, "MFMA only (12 per iter)");
    run<1, 12, 112>(W, out, cyc, "VALU only (112 pk_fma per iter)");
    if (W >= 2) run<3, 12, 112>(W, out, cyc, "MFMA waves + VALU waves (split)");
  }
  exact_table(out, reinterpret_cast<unsigned*>(cyc));
  kinds_table(out);
  run_phases<12, 112>(out);
  run_phases<24, 112>(out);
  return 0;
}
