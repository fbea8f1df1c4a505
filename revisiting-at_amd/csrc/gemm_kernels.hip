// gemm_kernels.hip — bf16 GEMM with fused epilogues for gfx950 (MI355X):   D = epilogue( A[M,K] · B[N,K]^T )
//
// The pointwise convolutions / linears that have no fused-block kernel (/root/reference/models/convnext.py:42-46 at C >= 384 in
// the training pass and at C >= 512 everywhere; the stage downsample convolutions :76-83 as GEMMs; the qkv / proj / fc1 / fc2
// linears of the timm transformer block the reference's ViTs are built from, utils_architecture.py:271-301) ran in hipBLASLt
// with one-pass kernels around them (bias, GELU, layer scale + residual, GELU').  Here they are one kernel each, with the
// surrounding element-wise work in the epilogue:
//     EPI_BIAS       D = bf16(acc + b)                                              qkv, proj, downsample, plain input gradients
//     EPI_BIAS_GELU  D = bf16(GELU(z)), z = bf16(acc + b); optionally Zout = z      fc1 (+ the pre-activation for the backward)
//     EPI_SCALE_RES  Y = bf16(acc + b); D = R + gamma * Y (fp32 or bf16 D / R)      fc2 + layer scale + residual
//     EPI_GELU_GRAD  D = bf16(acc * GELU'(Z))                                       dHpre = (dO W2) * GELU'(Hpre)
// Both operands are K-contiguous (A: activation rows; B: the [out, in] weight of nn.Linear as stored, or a transposed copy for
// input gradients), which is the MFMA fragment order of v_mfma_f32_32x32x16_bf16 for BOTH operands: lane (row = l & 31,
// half = l >> 5) holds 8 consecutive k of its row.
//
// Decomposition.  Workgroup tiles 256 x 256 (N a multiple of 256 and >= 512 tiles), 256 x 192 (every N of the models is a
// multiple of 192 = 6 x 32) or 128 x 192 (small grids); 8 (4) wavefronts of 64 x BN/2 = 2 x 4 (2 x 3) MFMA blocks, 128 (96)
// accumulator registers; K in steps of 64.  The kernel is bound by the rate at which a CU takes operand bytes from L2 (~15 B /
// cycle, profiles/r03_gemm.md), so the FLOP per staged byte of the tile (128 at 256 x 256, 110 at 256 x 192) is what counts.
// Staging: global_load_lds_dwordx4 (1 KiB = 8 rows x 128 B per instruction, no registers) into a two-stage ring, stage t + 1 in
// flight while stage t is consumed (its DMA instructions dealt over the four k-steps of stage t), one barrier per K step.  LDS rows
// are 128 B, so a 32-row fragment read would put a ds_read_b128 lane group on two 16-byte slots; the 16-byte chunk index is
// XOR-ed with (row >> 1) & 7 - applied to the per-lane SOURCE address of the DMA (its LDS destination is lane-linear) and to the
// read address - which spreads the 16 rows of a lane group over all 16 slots of the 256-byte bank row.
// Tile order: XCD-aware (block b runs on XCD b % 8: XCD x works through the x-th contiguous eighth of the tile list) and in
// panels of a few column tiles, so that the B rows an XCD keeps re-reading stay in its 4 MB L2 (see the kernel).
// Epilogue: the accumulators go through the (dead) staging ring so that every lane stores 16 contiguous bytes of an output row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "apgd_hip.h"
#include "convnext_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf16_round(float v) {            // round to nearest even, as a store to bf16 and back
  const uint32_t w = pack_bf16(v, 0.f);
  return bf16_lo(w);
}

// exact-erf GELU / GELU' (the polynomials of block_kernels.hip: |error| 1.2e-6 / 1.6e-5, tools/fit_gelu.py, fit_gelu_grad.py)
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
  return fmaf(az, fmaf(erfc_q(az), -0.5f, 0.5f), 0.5f * z);
}
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float a = fminf(fabsf(z), 6.0f);
  const float x = a * 0.8493218002880191f;
  const float E = __builtin_amdgcn_exp2f(-(x * x));
  float w = fmaf(1.8761737253e-03f, x, -1.8196647143e-02f);
  w = fmaf(w, x, 7.6242087502e-02f);
  w = fmaf(w, x, -1.9087504279e-01f);
  w = fmaf(w, x, 3.3884271219e-01f);
  w = fmaf(w, x, -9.3857446811e-01f);
  w = fmaf(w, x, 4.9998430368e-01f);
  return 0.5f + copysignf(0.5f - E * w, z);
}

// one global_load_lds_dwordx4: 64 lanes x 16 bytes -> 1 KiB of LDS at the wave-uniform address `dst` (see block_kernels.hip glds16)
// (M0 is set and left: nothing else in this kernel uses it)
__device__ __forceinline__ void glds16(const unsigned char* gsrc, uint32_t dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(dst) : "memory");
}

// (Measured and dropped in round 3, profiles/r03_gemm.md: operands staged through registers three stages deep on the same two LDS
//  buffers, 440 - 800 vs 480 - 870 TFLOP/s; a five-deep ring of 32-k stages: the depth of the prefetch does not bound this kernel.)

// 1: the DMA instructions of stage t + 1 are dealt over the four k-steps of stage t instead of being issued in one burst behind the
// barrier (every CU of the chip bursts at about the same time: the L2s see 56 - 64 KB requests per CU, then nothing)
#ifndef GEMM_SPREAD
#define GEMM_SPREAD 1
#endif

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_SCALE_RES = 2, EPI_GELU_GRAD = 3 };

struct GemmArgs {
  const uint16_t* A; long lda;     // [M, K] bf16, row stride lda elements
  const uint16_t* B; long ldb;     // [N, K] bf16
  void* D; long ldd;               // [M, N] bf16 (fp32 allowed for EPI_SCALE_RES)
  const float* bias;               // [N] or NULL
  const float* gamma;              // EPI_SCALE_RES: [N] or NULL (= 1)
  const void* R; long ldr;         // EPI_SCALE_RES: residual [M, N]
  uint16_t* Zout; long ldz;        // EPI_BIAS_GELU: pre-activation out (or NULL); EPI_SCALE_RES: Y = bf16(acc + b) out (or NULL)
  const uint16_t* Zin;             // EPI_GELU_GRAD: pre-activation [M, N] (row stride ldz)
  long M; int N, K;
  int pw;                          // column tiles per panel of the tile order (see the kernel)
};

constexpr int BK = 64;

template <int BM, int BN, int WM>
struct Geo {
  static constexpr int WN = BN / 2, NI = WM / 32, NJ = WN / 32;              // wavefront tile WM x WN = NI x NJ MFMA blocks
  static constexpr int WAVES_M = BM / WM, WAVES = WAVES_M * (BN / WN), THREADS = WAVES * 64;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;                   // 1 KiB DMA instructions per stage
  static constexpr int INSTR = (A_INSTR + B_INSTR) / WAVES;                  // per wavefront
  static_assert((A_INSTR + B_INSTR) % WAVES == 0 && A_INSTR % WAVES == 0, "whole DMA instructions per wavefront, A before B");
  static constexpr int LDS = 2 * STAGE;
  // epilogue: 32 rows x WN columns fp32 per wavefront and pass
  static_assert(WAVES * 32 * WN * 4 <= LDS, "the epilogue tile reuses the staging ring");
  static_assert(LDS <= 160 * 1024, "LDS of a CU");
};

template <int BM, int BN, int WM, int EPI, typename TD, typename TR>
__global__ __launch_bounds__((Geo<BM, BN, WM>::THREADS), (Geo<BM, BN, WM>::WAVES == 8 ? 2 : 1)) void gemm_nt_kernel(const GemmArgs p) {
  using G = Geo<BM, BN, WM>;
  constexpr int WN = G::WN, NI = G::NI, NJ = G::NJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  [[maybe_unused]] const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)lds));
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % G::WAVES_M, wn = wave / G::WAVES_M;
  // tile order: column tiles of one row tile are neighbours in the grid, so the workgroups that share an A tile run together
  // ... and, since block b runs on XCD b % 8 (each with its own L2), on ONE XCD: XCD x works through the x-th contiguous eighth of
  // the tile list (bijective for any tile count)
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const long nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7;
  const long xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
  const long tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  // ... in PANELS of p.pw column tiles, row by row inside a panel: the B rows of a panel (pw x 192 x K bf16) are what an XCD's
  // 32 workgroups re-read for every row tile, and they have to stay in its 4 MB L2 next to the A rows streaming through - with
  // whole rows of column tiles (pw = N / 192) the 4.7 MB weight of a 3072 x 768 layer came back from the fabric for every pair of
  // row tiles (L2 hit rate 68 %)
  const long n_tiles_m = (p.M + BM - 1) / BM;
  const long tpp = n_tiles_m * p.pw, full = n_tiles_n / p.pw;
  long trow; int tcol;
  {
    const long pn = tile / tpp;
    if (pn < full) {
      const long w = tile - pn * tpp;
      trow = w / p.pw;
      tcol = static_cast<int>(pn * p.pw + (w - trow * p.pw));
    } else {
      const int wl = n_tiles_n - static_cast<int>(full) * p.pw;
      const long w = tile - full * tpp;
      trow = w / wl;
      tcol = static_cast<int>(full * p.pw + (w - trow * wl));
    }
  }
  const long m0 = trow * BM;
  const int n0 = tcol * BN;

  // ---- DMA source addresses of this lane: instruction q of a stage covers tile rows 8 * (q * WAVES + wave) ... + 7 of A (then B)
  const int r8 = lane >> 3, sl = lane & 7;                                   // row inside the 8-row group, LDS chunk inside the row
  const unsigned char* src[G::INSTR];                                        // swizzled source chunk of the LDS-DMA
#pragma unroll
  for (int q = 0; q < G::INSTR; ++q) {
    const int j = q * G::WAVES + wave;                                       // 8-row group of the stage
    const bool isA = j < G::A_INSTR;
    const int row = (isA ? j : j - G::A_INSTR) * 8 + r8;                     // row inside the A / B tile
    const int chunk = sl ^ ((row >> 1) & 7);                                 // swizzle on the SOURCE (the destination is lane-linear)
    if (isA) {
      long gr = m0 + row;
      if (gr >= p.M) gr = p.M - 1;
      src[q] = reinterpret_cast<const unsigned char*>(p.A + gr * p.lda) + chunk * 16;
    } else {
      int gr = n0 + row;
      if (gr >= p.N) gr = p.N - 1;
      src[q] = reinterpret_cast<const unsigned char*>(p.B + static_cast<long>(gr) * p.ldb) + chunk * 16;
    }
  }
#define STAGE_LOAD(T)                                                                                          \
  {                                                                                                            \
    const uint32_t sb_ = lds0 + ((T) & 1) * G::STAGE;                                                          \
    _Pragma("unroll") for (int q = 0; q < G::INSTR; ++q)                                                       \
      glds16(src[q] + static_cast<long>(T) * (BK * 2), sb_ + (q * G::WAVES + wave) * 1024);                    \
  }

  // ---- fragment read offsets: chunk (2 ks + half) ^ ((row >> 1) & 7) of row l32 (+ 32-row block offsets, multiples of 16 rows)
  int foff[4];
  {
    const int t = half ^ ((l32 >> 1) & 7);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = l32 * 128 + (((2 * ks) ^ t) * 16);
  }
  const int a_base = wm * (WM * 128), b_base = G::A_BYTES + wn * (WN * 128);

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = p.K / BK;
  // one K step = 4 k-steps of 6 MFMAs; the 5 fragments of k-step ks + 1 are read while the MFMAs of k-step ks run (two register
  // sets, the order pinned: MFMA, read, MFMA, read, ...) - left to itself the compiler issued each k-step's reads right in
  // front of its first MFMA and the LDS latency (~130 cycles) was exposed four times per K step
#define FRAG_READ(AF, BF, KS)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) AF[i] = *reinterpret_cast<const bf16x8*>(sb_ + a_base + i * 4096 + foff[KS]); \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) BF[j] = *reinterpret_cast<const bf16x8*>(sb_ + b_base + j * 4096 + foff[KS]);
#define STAGE_LOAD_PART(T, KS)                                                                                 \
  {                                                                                                            \
    const uint32_t sl_ = lds0 + ((T) & 1) * G::STAGE;                                                          \
    _Pragma("unroll") for (int q = (KS) * G::INSTR / 4; q < ((KS) + 1) * G::INSTR / 4; ++q)                    \
      glds16(src[q] + static_cast<long>(T) * (BK * 2), sl_ + (q * G::WAVES + wave) * 1024);                    \
  }
#define KSTEP(AF, BF, AN, BN_, KS)                                                                             \
    if (GEMM_SPREAD && pre_) STAGE_LOAD_PART(tn_, KS)                                                          \
    if ((KS) < 3) { FRAG_READ(AN, BN_, ((KS) < 3 ? (KS) + 1 : 3)) }                                            \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                             \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], acc[i][j], 0, 0, 0);                 \
    if ((KS) < 3) {                                                                                            \
      _Pragma("unroll") for (int q_ = 0; q_ < NI + NJ; ++q_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
      }                                                                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, NI * NJ - NI - NJ, 0);                                       \
    }
#define COMPUTE(T, PRE)                                                                                        \
  {                                                                                                            \
    const unsigned char* sb_ = lds + ((T) & 1) * G::STAGE;                                                     \
    [[maybe_unused]] const bool pre_ = (PRE);                                                                  \
    [[maybe_unused]] const int tn_ = (T) + 1;                                                                  \
    bf16x8 fa0[NI], fb0[NJ], fa1[NI], fb1[NJ];                                                                 \
    FRAG_READ(fa0, fb0, 0)                                                                                     \
    KSTEP(fa0, fb0, fa1, fb1, 0)                                                                               \
    KSTEP(fa1, fb1, fa0, fb0, 1)                                                                               \
    KSTEP(fa0, fb0, fa1, fb1, 2)                                                                               \
    KSTEP(fa1, fb1, fa0, fb0, 3)                                                                               \
  }
  STAGE_LOAD(0)
  for (int t = 0; t + 1 < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // stage t has landed (this wavefront's pieces)
    __builtin_amdgcn_s_barrier();                                          // ... everyone's; and nobody reads stage t - 1 any more
#if GEMM_SPREAD
    COMPUTE(t, true)
#else
    STAGE_LOAD(t + 1)
    COMPUTE(t, false)
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  COMPUTE(nt - 1, false)
#undef COMPUTE
#undef KSTEP
#undef STAGE_LOAD_PART
#undef FRAG_READ
#undef STAGE_LOAD
  __syncthreads();                                                          // the ring is dead: epilogue scratch

  // ---- epilogue.  acc[i][j][r] = C[wm*WM + i*32 + (r&3) + 8*(r>>2) + 4*half][wn*WN + j*32 + l32].  NI passes (i = 0, 1, ..) of 32
  //      rows x WN columns through 32 x WN x 4 bytes of LDS per wavefront (fp32), read back as 4-column chunks of a row (WN / 4
  //      per row, WN / 8 per lane and pass) - a row's chunks sit in consecutive lanes, so loads and stores are contiguous runs.
  float* scr = reinterpret_cast<float*>(lds) + wave * (32 * WN);
  const int ncol0 = n0 + wn * WN;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * WN + j * 32 + l32] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    const long mrow0 = m0 + wm * WM + i * 32;
#pragma unroll
    for (int c = 0; c < WN / 8; ++c) {                                       // 32 x WN / 4 chunks per pass, 64 lanes
      const int idx = c * 64 + lane, rr = idx / (WN / 4), c4 = (idx - rr * (WN / 4)) * 4;
      const long m = mrow0 + rr;
      const int n = ncol0 + c4;
      if (m >= p.M || n >= p.N) continue;                                    // (N is a multiple of 4: whole chunks)
      const float4 v = *reinterpret_cast<const float4*>(scr + rr * WN + c4);
      float y[4] = {v.x, v.y, v.z, v.w};
      if (EPI != EPI_GELU_GRAD && p.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        y[0] += bf16_round(b4.x); y[1] += bf16_round(b4.y); y[2] += bf16_round(b4.z); y[3] += bf16_round(b4.w);
      }
      if constexpr (EPI == EPI_BIAS) {
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + m * p.ldd + n) = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
      } else if constexpr (EPI == EPI_BIAS_GELU) {
        const uint2 z = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
        if (p.Zout) *reinterpret_cast<uint2*>(p.Zout + m * p.ldz + n) = z;
        const float g0 = gelu_f(bf16_lo(z.x)), g1 = gelu_f(bf16_hi(z.x)), g2 = gelu_f(bf16_lo(z.y)), g3 = gelu_f(bf16_hi(z.y));
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + m * p.ldd + n) = make_uint2(pack_bf16(g0, g1), pack_bf16(g2, g3));
      } else if constexpr (EPI == EPI_SCALE_RES) {
        const uint2 z = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
        if (p.Zout) *reinterpret_cast<uint2*>(p.Zout + m * p.ldz + n) = z;
        float g[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.gamma) {
          const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n);
          g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
        }
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.R) {
          if constexpr (sizeof(TR) == 4) {
            const float4 r4 = *reinterpret_cast<const float4*>(static_cast<const float*>(p.R) + m * p.ldr + n);
            rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
          } else {
            const uint2 r2 = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.R) + m * p.ldr + n);
            rv[0] = bf16_lo(r2.x); rv[1] = bf16_hi(r2.x); rv[2] = bf16_lo(r2.y); rv[3] = bf16_hi(r2.y);
          }
        }
        const float o0 = fmaf(bf16_lo(z.x), g[0], rv[0]), o1 = fmaf(bf16_hi(z.x), g[1], rv[1]);
        const float o2 = fmaf(bf16_lo(z.y), g[2], rv[2]), o3 = fmaf(bf16_hi(z.y), g[3], rv[3]);
        if constexpr (sizeof(TD) == 4)
          *reinterpret_cast<float4*>(static_cast<float*>(p.D) + m * p.ldd + n) = make_float4(o0, o1, o2, o3);
        else
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + m * p.ldd + n) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
      } else {                                                               // EPI_GELU_GRAD
        const uint2 z = *reinterpret_cast<const uint2*>(p.Zin + m * p.ldz + n);
        // dH is a bf16 tensor in the library composition (the GEMM's output) before it meets GELU': round first
        const float d0 = bf16_round(y[0]) * gelu_grad_f(bf16_lo(z.x)), d1 = bf16_round(y[1]) * gelu_grad_f(bf16_hi(z.x));
        const float d2 = bf16_round(y[2]) * gelu_grad_f(bf16_lo(z.y)), d3 = bf16_round(y[3]) * gelu_grad_f(bf16_hi(z.y));
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + m * p.ldd + n) = make_uint2(pack_bf16(d0, d1), pack_bf16(d2, d3));
      }
    }
  }
}

template <int BM, int BN, int WM, int EPI, typename TD, typename TR>
int launch(const GemmArgs& a, hipStream_t s) {
  using G = Geo<BM, BN, WM>;
  auto kfn = gemm_nt_kernel<BM, BN, WM, EPI, TD, TR>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr_done = true;
  }
  const long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kfn, dim3(static_cast<unsigned>(tiles)), dim3(G::THREADS), G::LDS, s, a);
  return launch_status();
}

int g_tile_force = 0;

template <int BM, int BN, int WM>
int dispatch(const GemmArgs& a, int epi, int d_dtype, int r_dtype, hipStream_t s) {
  switch (epi) {
    case EPI_BIAS: return launch<BM, BN, WM, EPI_BIAS, uint16_t, uint16_t>(a, s);
    case EPI_BIAS_GELU: return launch<BM, BN, WM, EPI_BIAS_GELU, uint16_t, uint16_t>(a, s);
    case EPI_GELU_GRAD: return launch<BM, BN, WM, EPI_GELU_GRAD, uint16_t, uint16_t>(a, s);
    case EPI_SCALE_RES:
      if (d_dtype == APGD_F32) return r_dtype == APGD_F32 ? launch<BM, BN, WM, EPI_SCALE_RES, float, float>(a, s)
                                                          : launch<BM, BN, WM, EPI_SCALE_RES, float, uint16_t>(a, s);
      return r_dtype == APGD_F32 ? launch<BM, BN, WM, EPI_SCALE_RES, uint16_t, float>(a, s)
                                 : launch<BM, BN, WM, EPI_SCALE_RES, uint16_t, uint16_t>(a, s);
    default: return APGD_ERR_ARG;
  }
}

}  // namespace

int gemm_nt_tile_switch(int value) {
  const int prev = g_tile_force;
  if (value >= 0) g_tile_force = value & 3;
  return prev;
}

extern "C" {

int cnx_gemm_nt_supported(int64_t M, int32_t N, int32_t K) { return (M > 0 && N > 0 && N % 4 == 0 && K > 0 && K % 64 == 0) ? 1 : 0; }

int cnx_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* D, int64_t ldd, int d_dtype, int64_t M, int32_t N,
                int32_t K, int32_t epilogue, const float* bias, const float* gamma, const void* R, int64_t ldr, int r_dtype,
                void* z_out, const void* z_in, int64_t ldz, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return APGD_ERR_SIZE;
  if (M == 0) return APGD_OK;
  if (!A || !B || !D) return APGD_ERR_NULL;
  if (!cnx_gemm_nt_supported(M, N, K)) return APGD_ERR_ARG;
  if (lda < K || ldb < K || ldd < N || (lda & 7) || (ldb & 7) || (ldd & 3)) return APGD_ERR_ARG;      // 16-byte rows / chunks
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(D)) & 15) return APGD_ERR_ARG;
  if (epilogue == EPI_SCALE_RES) {
    if ((d_dtype != APGD_F32 && d_dtype != APGD_BF16) || (R && r_dtype != APGD_F32 && r_dtype != APGD_BF16)) return APGD_ERR_DTYPE;
    if (R && (ldr < N || (ldr & 3))) return APGD_ERR_ARG;
  } else if (d_dtype != APGD_BF16) {
    return APGD_ERR_DTYPE;
  }
  if (epilogue == EPI_GELU_GRAD && !z_in) return APGD_ERR_NULL;
  if ((z_out || z_in) && (ldz < N || (ldz & 3))) return APGD_ERR_ARG;
  GemmArgs a;
  a.A = static_cast<const uint16_t*>(A); a.lda = lda; a.B = static_cast<const uint16_t*>(B); a.ldb = ldb;
  a.D = D; a.ldd = ldd; a.bias = bias; a.gamma = gamma; a.R = R; a.ldr = ldr;
  a.Zout = static_cast<uint16_t*>(z_out); a.Zin = static_cast<const uint16_t*>(z_in); a.ldz = ldz;
  a.M = M; a.N = N; a.K = K;
  // tile: 256 x 256 (128 FLOP per staged byte) where N is a multiple of 256 and the grid still has >= ~2 rounds of workgroups;
  // else 256 x 192 (every N of the models is a multiple of 192) when that gives >= ~3 rounds; else 128 x 192
  // (cnx_runtime_switch(CNX_SWITCH_GEMM_NT_TILE): 0 = this rule; 1 / 2 / 3 force 128 x 192 / 256 x 192 / 256 x 256 where N allows - A/B runs)
  const int force = g_tile_force;
  const int bm_env = force == 1 ? 128 : force >= 2 ? 256 : 0;
  const int bn_env = force == 3 && N % 256 == 0 ? 256 : force ? 192 : 0;
  constexpr int pw_env = 0;
  const long rows256 = (M + 255) / 256;
  int bn = (N % 256 == 0 && rows256 * (N / 256) >= 512) ? 256 : 192;
  if (bn_env == 192 || bn_env == 256) bn = bn_env;
  const int ntn = (N + bn - 1) / bn;
  const bool big = bn == 256 || (bm_env ? bm_env == 256 : rows256 * ntn >= 768);
  // panel width: the widest whose B rows (pw x BN x K bf16) take at most ~1.5 MB of an XCD's 4 MB L2
  int pw = static_cast<int>((1536L * 1024) / (static_cast<long>(bn) * K * 2));
  if (pw_env > 0) pw = pw_env;
  a.pw = pw < 1 ? 1 : (pw > ntn ? ntn : pw);
  // (the wavefront-tile height is a template parameter: 256 x 256 on FOUR wavefronts of 128 x 128 - 256 accumulator registers,
  //  half the fragment reads per MFMA, one wavefront per SIMD - measured 25 - 40 % slower than eight of 64 x 128)
  if (bn == 256) return dispatch<256, 256, 64>(a, epilogue, d_dtype, r_dtype, as_stream(stream));
  if (big) return dispatch<256, 192, 64>(a, epilogue, d_dtype, r_dtype, as_stream(stream));
  return dispatch<128, 192, 64>(a, epilogue, d_dtype, r_dtype, as_stream(stream));
}

}  // extern "C"
