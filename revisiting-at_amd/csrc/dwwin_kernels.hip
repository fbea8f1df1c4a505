// dwwin_kernels.hip — depthwise 7x7 on bf16 operands with a REGISTER sliding window (round 4; gfx950 / MI355X).
// Reference arithmetic: /root/reference/models/convnext.py:28, 39 (`Conv2d(dim, dim, 7, padding=3, groups=dim)` under autocast:
// bf16 inputs and weights, fp32 accumulation), forward and - with the 180-degree rotated filter - its input gradient, with the
// residual gradient added in the same pass; and the filter / bias gradient on the same window.
//
// Why another form.  The LDS-ring kernels of model_kernels.hip (dwconv7x7_roll / _multi) stage every input row through
// registers into a pair-packed LDS ring and back, in workgroup-wide phases separated by barriers; with their prefetch registers
// they sit at 220 - 256 VGPRs (+ up to 66 AGPRs): ONE or TWO wavefronts per SIMD, 4 - 8 per CU.  Round 3 measured them at half
// of both of their bounds (56x56x96 forward 145 us against 73 us of HBM time and ~60 us of v_dot2 time).
//
// Here a wavefront is on its own - no barrier, nothing shared between wavefronts:
//   * lane = (unit, channel): CH = 32 or 64 channels of one STRIP of 7 output columns ("unit" = strip of one image, a wavefront
//     holds 64 / CH units; 56, 28, 14 and 7 - every ConvNeXt map at 224 - are multiples of 7, other widths end in a ragged strip).
//   * the lane keeps a 7-row x 13-column window of ITS channel in 49 registers, packed as pairs of W-adjacent bf16 per dword
//     (columns 7 s - 3 ... 7 s + 9 of strip s), walks DOWN a band of the image, and per output row takes in the row that enters
//     the window, runs 7 x 28 v_dot2_f32_bf16 on the window and puts out the 7 values of the finished row:
//         even t = 2 q  : pairs q .. q+3 . {(f0,f1),(f2,f3),(f4,f5),(f6,0)}
//         odd  t = 2 q+1: pairs q .. q+3 . {(0,f0),(f1,f2),(f3,f4),(f5,f6)}            (as the LDS kernels; no realignment)
// The price is the column halo (13 columns in per 7 out: neighbouring strips sit in the same wavefront / workgroup, the re-reads
// are L1 / L2 hits) and the 6 halo rows per band (XCD-aware item order keeps the bands of an image in one L2).
//
// Algorithmic bytes: forward fp32 -> bf16: 6 B per element; input gradient bf16 -> fp32 with the fp32
// residual gradient added: 10 B per element.  HBM-bound: 56 x 56 x 96, batch 256: 462 / 771 MB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "apgd_hip.h"
#include "convnext_hip.h"
#include "dw_internal.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// compile-time ablations of the forward / input-gradient kernel (measurement builds only: `make EXTRA=-DDW_ABL=n`, profiles/r05_dw.md):
// 1 = no dot products of filter rows 0..5, 2 = no row / add-row DMA in the steps, 4 = no output staging and no stores
// (the address-pattern variants of profiles/r05_dw.md section 4 - contiguous stores / loads, loads into registers - were timing-only
//  builds with wrong results; they are in the history of this file, not in it)
#ifndef DW_ABL
#define DW_ABL 0
#endif

constexpr int kT = 7;                 // output columns of a strip
constexpr int kCols = kT + 6;         // input columns a strip reads (13)
constexpr int kPairs = 7;             // window dwords per row: 14 columns, the last one always zero

template <typename TI> struct RawRow { TI v[kCols]; };

__device__ __forceinline__ void pack_row(uint32_t (&d)[kPairs], const RawRow<float>& r, const uint32_t (&m)[kPairs]) {
#pragma unroll
  for (int i = 0; i < kPairs; ++i) d[i] = pack2_bf16(r.v[2 * i], 2 * i + 1 < kCols ? r.v[2 * i + 1 < kCols ? 2 * i + 1 : 0] : 0.f) & m[i];
}
__device__ __forceinline__ void pack_row(uint32_t (&d)[kPairs], const RawRow<uint16_t>& r, const uint32_t (&m)[kPairs]) {
#pragma unroll
  for (int i = 0; i < kPairs; ++i) {
    const uint32_t hi = 2 * i + 1 < kCols ? static_cast<uint32_t>(r.v[2 * i + 1 < kCols ? 2 * i + 1 : 0]) : 0u;
    d[i] = (static_cast<uint32_t>(r.v[2 * i]) | (hi << 16)) & m[i];
  }
}

struct WinArgs {
  const void* x; const float* w49c; const float* bias; const float* add; void* out;
  int N, H, W, C, flip;
  int n_strips, n_sg, n_cg, band, n_bands;   // strips per image row, strip groups (of 64 / CH strips), channel groups, rows per band, bands
  long items_per_cg, items_per_cg_real;      // wavefront work items of one channel group: N * n_bands * n_sg, padded to a multiple of 4
};

// ------------------------------------------------------------------------------------------------------------------------
// Forward / input gradient.  One wavefront = one item: (image, band, channel group, strip group), strip group fastest.  Every
// global access goes THROUGH LDS: rows in by DMA (`buffer_load_dwordx4 ... lds`), rows out as 16-byte stores.
//
// Why.  The first form of this kernel (round 4, `git log`: dwconv7x7_win_kernel) loaded a row's 13 values per lane straight into
// registers, one row ahead of its use, and stored 7 values per lane.  Its counters (profiles/r04_dwwin.md): 53 % of the wavefront
// cycles parked in s_waitcnt at 56 x 56 x 96 with the VALU 50 % busy - and 4.35 M VMEM instructions per launch, ~24 cycles of kernel time each per CU: 13 loads + 7 stores per output
// row, every one of them moving 4 (or 2) bytes per lane.  The texture path takes an instruction's 64 addresses at a fixed rate
// whatever their width, so a lane = (strip, channel) kernel that loads one element per lane and instruction is bound by the NUMBER
// of its memory instructions.  Two steps, both measured:
//   1. the row that enters the window comes in by LDS-DMA, two steps ahead of its use (K = 2 row slots per wavefront; an LDS-DMA load
//      has no destination register, so the depth of the prefetch costs LDS, not VGPRs: 13 in-flight registers per row had capped
//      the register form at one row ahead).  With one dword per lane and instruction: 170 -> 144 us (fp32 forward, 56 x 56 x 96).
//   2. one instruction moves 16 bytes per lane: a row of a strip group is 13 column runs of 256 (fp32) / 128 (bf16) contiguous
//      bytes; lane l of a load fetches chunk l % LPC of column l / LPC, and because LDS-DMA writes lane l at dst + 16 l the slot comes
//      out as [column][(unit, channel)] - the lane = (unit, channel) layout the window reads with ds_read_b32 / _u16.  4 (fp32) or
//      2 (bf16) loads per row instead of 13, 2 for the add operand's row instead of 7; the finished row goes the other way: 7
//      ds_write into [column][(unit, channel)], one ds_read_b128 and ONE (bf16) or TWO (fp32) buffer_store_dwordx4 instead of 7
//      stores.  5 - 6 VMEM instructions per output row instead of 20 - 27.
// With loads and stores out of the register file the kernel needs ~75 VGPRs; the packed filter (56 dwords per lane) moves from LDS
// into registers (no 14 KiB of LDS reads per wavefront and row), ~130 VGPRs, three wavefronts per SIMD.
// A band of an image pays three full memory latencies for its first six rows (nothing to overlap them with), so the launcher cuts
// the images into as few bands as fill the chip ONCE (3072 wavefronts = every ConvNeXt-T stage at batch 256 in one band).
// Measured, batch 256, us (this form / register form / LDS-ring kernels): 56 x 56 x 96 forward 128 / 171 / 156, input gradient + add
// 172 / 196 / 197;  28 x 28 x 192 61 / 71 / 113, 85 / 94 / 137;  14 x 14 x 384 37 / 42 / 49, 40 / 47 / 63;  7 x 7 x 768 25 / 26 / 35; the step
// 49.5 - 50.0 ms against 51.0 - 51.7 (profiles/r04_dwwin.md).
//
// Counting.  The DMA loads are inline asm (the builtin form makes the compiler's wait-count pass wait for ZERO outstanding
// loads in front of every LDS read it cannot tell apart from the ring: block_kernels.hip, glds16); the compiler sees no VMEM load
// at all in the row loop - only its stores, which need no wait - so every vmcnt wait below is ours, and the number of VMEM
// instructions per step is a compile-time constant: [NA add-row DMA] NR row DMA ... NSO stores - a step that has no row to store (the
// surplus steps of a band's last group) issues NSO sink loads instead; a half-idle wavefront (odd strip count at 32-channel groups)
// stores under EXEC.  (Loads and stores retire vmcnt in issue order on this family: the compiler's own counted waits rely on it.)
// The ring is wave-private: a row is ordered for its reader by the issuing wavefront's own counted vmcnt, a slot is re-issued
// behind the lgkmcnt(0) that retired its reads.
typedef __attribute__((ext_vector_type(4))) uint32_t rsrc4_t;

template <int BYTES>
__device__ __forceinline__ void dma_lds(uint32_t lds_dst, uint32_t voff, rsrc4_t rs, uint32_t soff) {
  // lds_dst: wave-uniform LDS byte address (M0); lane l's dword lands at lds_dst + 4 l (zero-extended for the 16-bit form)
  if constexpr (BYTES == 4)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ushort %1, %2, %3 offen lds" : : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// 16 bytes per lane: one instruction moves 1 KiB (four 256-byte or eight 128-byte column runs of a row); lane l lands at lds_dst + 16 l
__device__ __forceinline__ void dma_lds16(uint32_t lds_dst, uint32_t voff, rsrc4_t rs, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// the same under a wave-uniform lane mask: only the lanes of `mask` fetch (and land at lds_dst + 16 l); one VMEM instruction whatever the mask
__device__ __forceinline__ void dma_lds16_masked(uint32_t lds_dst, uint32_t voff, rsrc4_t rs, uint32_t soff, uint64_t mask) {
  uint64_t sv;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
               : "=&s"(sv) : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff), "s"(mask) : "memory");
}

__device__ __forceinline__ rsrc4_t make_rsrc4(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  rsrc4_t r;
  r.x = static_cast<uint32_t>(a);
  r.y = static_cast<uint32_t>(a >> 32) & 0xffffu;          // stride 0: raw buffer
  r.z = bytes;
  r.w = 0x00020000u;                                       // as __builtin_amdgcn_make_buffer_rsrc(..., 0x00020000) above
  return r;
}

__device__ __forceinline__ void pack_row_lds(uint32_t (&d)[kPairs], const uint32_t (&r)[kCols], const uint32_t (&m)[kPairs], float) {
#pragma unroll
  for (int i = 0; i < kPairs; ++i)
    d[i] = pack2_bf16(__uint_as_float(r[2 * i]), 2 * i + 1 < kCols ? __uint_as_float(r[2 * i + 1 < kCols ? 2 * i + 1 : 0]) : 0.f) & m[i];
}
__device__ __forceinline__ void pack_row_lds(uint32_t (&d)[kPairs], const uint32_t (&r)[kCols], const uint32_t (&m)[kPairs], uint16_t) {
#pragma unroll
  for (int i = 0; i < kPairs; ++i) {
    const uint32_t hi = 2 * i + 1 < kCols ? r[2 * i + 1 < kCols ? 2 * i + 1 : 0] : 0u;
    d[i] = ((r[2 * i] & 0xffffu) | (hi << 16)) & m[i];
  }
}

// SH (round 5, CH = 32 only): the two strips of a wavefront are NEIGHBOURS - columns 14 g - 3 .. 14 g + 16 of strip pair g - and
// share their column halo: a slot is [20 columns][32 channels] (2.5 KiB fp32 / 1.25 KiB bf16) instead of [13 (16) columns][2 units x
// 32 channels] (4 / 2 KiB), the add row and the output row are 14 contiguous columns x 32 channels (1.75 KiB instead of 2).  Unit u
// of the wavefront reads slot columns 7 u .. 7 u + 12: at 128 (64) bytes per column the two units are 896 (448) bytes apart = 32 (48)
// banks, so the 64 lanes of a ds_read still cover disjoint banks.  Why it pays: per output row a CU's 12 wavefronts took in
// 12 x (4 + 0.875) KiB = 21 bytes per cycle at the VALU's pace (2880 cycles per row step) - the L2 -> LDS path of a CU delivers
// 12 - 15, so the kernel ran at the ingest rate, not the VALU's (58 % busy); the shared halo asks for 14.5.
//
// K / AD (round 5): depth of the prefetch, a template parameter.  The row of step s + K is issued in step s (K ring slots per wavefront),
// the add row of step s + AD at the top of step s (AD + 1 slots); LDS-DMA has no destination registers, so depth is paid in LDS only and
// the slots are trimmed to the bytes that land.  Built because the ablations (profiles/r05_dw.md) show the kernel at the pace of its memory
// pipeline (fp32 forward, 56 x 56 x 96: 130 us; 128 with 86 % of the dot products removed; 94 with loads AND stores removed, 100 / 102 with
// either) - and measured flat: K = 2 .. 4, AD = 0 .. 3 all within the run-to-run spread.  What did move it is the ORDER of the items (below).
template <typename TI, typename TO, int CH, bool ADD, bool SH, int K, int AD>
__global__ __launch_bounds__(256, 3)
void dwconv7x7_dma_kernel(const WinArgs a) {
  static_assert(!SH || CH == 32, "the shared halo is the two strips of a 32-channel wavefront");
  static_assert(K >= 2 && K <= 6 && AD >= 0 && AD <= 4, "prefetch depth");
  constexpr int UPW = 64 / CH, NS = 7;
  // a row slot: 16 columns (13 used) x 64 lanes x sizeof(TI), written by NR wide loads of CPI columns each (SH: 20 columns x 32 channels)
  constexpr int COLB = (SH ? 32 : 64) * sizeof(TI);                        // bytes of one column in a slot
  constexpr int CPI = 1024 / COLB;                                         // columns per 1 KiB instruction: 4 (fp32) / 8 (bf16); SH: 8 / 16
  constexpr int NCOL = SH ? 20 : 13;                                       // columns of a slot that are read
  constexpr int NR = SH ? (NCOL + CPI - 1) / CPI : 16 / CPI;               // row loads per step: 4 (fp32) / 2 (bf16); SH: 3 / 2
  constexpr int NA = 2;                                                    // add-row loads per step (8 columns x 256 bytes, 7 used; SH: 14 x 128)
  constexpr uint32_t SLOT = SH ? NCOL * COLB : NR * 1024u;               // (SH: the last load lands under a lane mask - nothing behind column 19)
  constexpr int CSTR = COLB / sizeof(TI);                                  // a column's stride in the slot, elements
  static_assert(SH || (sizeof(TI) == 4 && NR == 4) || (sizeof(TI) == 2 && NR == 2), "row loads");
  // vmcnt of the row wait: VMEM instructions issued after the DMA group of the row a step needs (issued two steps before it, behind
  // that step's add-row DMA): NSO stores, [NA add DMA,] NR row DMA, NSO stores
  // output row: staged through LDS (the lane's 7 values -> [column][(unit, channel)]) and stored as NSO instructions of 16 bytes per lane
  constexpr int COLBO = (SH ? 32 : 64) * sizeof(TO), CPIO = 1024 / COLBO, LPCO = COLBO / 16, LPUO = SH ? LPCO : LPCO / UPW, EPLO = 16 / sizeof(TO);
  constexpr int NOUT = SH ? 2 * kT : kT;                                   // output columns of a wavefront's row
  constexpr int NSO = (NOUT + CPIO - 1) / CPIO;                            // 1 (bf16) / 2 (fp32), either form
  constexpr int CSTRO = COLBO / sizeof(TO);
  // VMEM instructions of a step, in issue order: [NA add-row DMA] ... NR row DMA ... NSO stores.  Behind the DMA group of the row a step
  // needs (issued K steps before it): the rest of that step and K - 1 whole steps; behind the add row it needs (issued AD steps before it, at
  // the top): that step's row DMA and stores, AD - 1 whole steps, and its own add and row DMA
  constexpr int NAE = ADD ? NA : 0, PER_STEP = NAE + NR + NSO;
  constexpr int WAIT_ROW = NSO + (K - 1) * PER_STEP;
  constexpr int WAIT_ADD = AD == 0 ? NR : (NR + NSO) + (AD - 1) * PER_STEP + NAE + NR;
  static_assert(WAIT_ROW < 64 && WAIT_ADD < 64, "vmcnt is a 6-bit count");
  constexpr uint32_t ASLOT = SH ? 2u * kT * 128u : 2048u;                  // an add-row slot: 14 x 128 bytes / 8 x 256 bytes
  const int lane = threadIdx.x & 63;
  long blk;
  {
    const long L = blockIdx.x, B = gridDim.x;
    const long q = B / 8, r = B % 8, xcd = L % 8, k = L / 8;
    blk = xcd * q + (xcd < r ? xcd : r) + k;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long item = blk * 4 + wave;
  // item = (image, band, strip group, channel group), channel group FASTEST: the wavefronts that share a pixel's cache lines (its 96
  // channels are 192 bytes of bf16 - three 64-byte pieces, half lines) sit next to each other in one workgroup / on one XCD, whose L2 then
  // sees whole lines.  Round 4 had the channel group slowest - the pieces of a line went through three different L2s: 128 - 133 us for the
  // fp32 forward at 56 x 56 x 96; this order 112 - 118; strip group fastest, then the channel group 117 - 124 (profiles/r05_dw.md)
  const bool item_ok = item < a.items_per_cg_real * a.n_cg;
  const int cg = static_cast<int>(item % a.n_cg);
  const long it = item / a.n_cg;
  const int sg = static_cast<int>(it % a.n_sg);
  const int bd = static_cast<int>((it / a.n_sg) % a.n_bands);
  const long n = item_ok ? it / (static_cast<long>(a.n_sg) * a.n_bands) : 0;
  const int H = a.H, W = a.W, C = a.C;
  const int ul = lane / CH;
  const bool unit_ok = sg * UPW + ul < a.n_strips;
  const int ulc = (unit_ok || SH) ? ul : a.n_strips - 1 - sg * UPW;       // (SH: an idle unit's window lies beyond the row: all masks zero)
  const int w0 = (sg * UPW + ulc) * kT;
  const int c = cg * CH + (lane % CH);
  const int r_begin = bd * a.band, r_end = min(H, r_begin + a.band);
  const int n_rows = r_end - r_begin;
  const long rs = static_cast<long>(W) * C;

  uint32_t m[kPairs];
#pragma unroll
  for (int i = 0; i < kPairs; ++i) {
    const int wl_ = w0 - 3 + 2 * i, wh = wl_ + 1;
    m[i] = ((wl_ >= 0 && wl_ < W) ? 0x0000ffffu : 0u) | ((2 * i + 1 < kCols && wh >= 0 && wh < W) ? 0xffff0000u : 0u);
  }
  const uint32_t zero_m[kPairs] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};

  const long goff = static_cast<long>(sg * UPW * kT - 3) * C + cg * CH;
  const long last_row = static_cast<long>(a.N) * H - 1;
  const uint32_t tensor_elems = static_cast<uint32_t>(static_cast<long>(a.N) * H * rs);
  const rsrc4_t rx = make_rsrc4(a.x, tensor_elems * static_cast<uint32_t>(sizeof(TI)));
  const rsrc4_t ra = make_rsrc4(a.add, ADD ? tensor_elems * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, tensor_elems * static_cast<uint32_t>(sizeof(TO)), 0x00020000);
  const uint32_t img_elem = static_cast<uint32_t>(n * H * rs);

  __shared__ __attribute__((aligned(16))) unsigned char ring[4][K][SLOT];
  __shared__ __attribute__((aligned(16))) unsigned char aring[ADD ? 4 : 1][ADD ? AD + 1 : 1][ADD ? ASLOT : 16];   // add rows of steps s .. s + AD
  __shared__ uint32_t sink[64];                                           // target of the loads that only keep the count (never read)
  __shared__ __attribute__((aligned(16))) unsigned char stg[4][(SH ? 16 : 8) * COLBO]; // output row of the step on its way to 16-byte stores
  if (!item_ok) return;
  // ---- the lane's packed filter in 56 registers (the register form above keeps it in LDS: with 13 load destinations and 7 store
  //      operands per row there was no room; here loads and stores go through LDS and the per-step filter reads - 14 KiB of LDS
  //      traffic per wavefront and row - are gone):  we[kh] . pairs (q .. q+3) for even t = 2q, wo[kh] for odd t
  uint32_t we[7][4], wo[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    float f[7];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int tap = kh * 7 + kw;
      f[kw] = a.w49c[(a.flip ? 48 - tap : tap) * C + c];
    }
    we[kh][0] = pack2_bf16(f[0], f[1]); we[kh][1] = pack2_bf16(f[2], f[3]); we[kh][2] = pack2_bf16(f[4], f[5]); we[kh][3] = pack2_bf16(f[6], 0.f);
    wo[kh][0] = pack2_bf16(0.f, f[0]); wo[kh][1] = pack2_bf16(f[1], f[2]); wo[kh][2] = pack2_bf16(f[3], f[4]); wo[kh][3] = pack2_bf16(f[5], f[6]);
  }
  typedef const volatile __attribute__((address_space(3))) uint32_t* lds_u32_t;
  typedef const volatile __attribute__((address_space(3))) uint16_t* lds_u16_t;
  typedef volatile __attribute__((address_space(3))) uint32_t* lds_w32_t;
  typedef volatile __attribute__((address_space(3))) uint16_t* lds_w16_t;
  typedef const volatile __attribute__((address_space(3))) u32x4_t* lds_v4_t;
  const uint32_t ring_w = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)&ring[wave][0][0]));
  const uint32_t aring_w = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)&aring[ADD ? wave : 0][0][0]));
  const uint32_t sink_w = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)&sink[0]));
  // ---- per-lane source offsets of the wide loads.  A 1 KiB load covers CPI columns; inside a column the 64 (unit, channel) values
  //      are 256 / 128 contiguous bytes of LDS but UPW separate runs of memory (a unit's strip starts 7 columns after its
  //      neighbour's): lane l -> column l / LPC, unit (l % LPC) / LPU, 16-byte chunk l % LPU of the unit's CH-channel run.
  //      SH: lane l -> column l / LPC of the pair's 20, 16-byte chunk l % LPC of its 32-channel run; the last load runs under a lane mask.
  constexpr int LPC = COLB / 16, LPU = SH ? LPC : LPC / UPW, EPL = 16 / sizeof(TI);   // lanes per column, per unit run; elements per lane
  const int wcol = lane / LPC, wun = SH ? 0 : (lane % LPC) / LPU, wch = lane % LPU;
  const int wun_c = (SH || sg * UPW + wun < a.n_strips) ? wun : a.n_strips - 1 - sg * UPW;  // an idle unit shadows the last strip
  const uint32_t vw_in = static_cast<uint32_t>((wun_c * kT + wcol) * C + wch * EPL) * static_cast<uint32_t>(sizeof(TI));
  // the last load of a row starts at column (NR - 1) CPI: its lanes beyond column 12 re-read column 12 (valid memory, unused slot columns)
  const int wcol_l = min(wcol, NCOL - 1 - (NR - 1) * CPI);
  const uint32_t vw_in_l = static_cast<uint32_t>((wun_c * kT + wcol_l) * C + wch * EPL) * static_cast<uint32_t>(sizeof(TI));
  constexpr uint64_t MASK_IN_L = SH ? ((NCOL - (NR - 1) * CPI) * LPC >= 64 ? ~0ull : (1ull << ((NCOL - (NR - 1) * CPI) * LPC)) - 1ull) : ~0ull;
  // output row: lane l of store k -> column k CPIO + l / LPCO (columns > 6: no store), unit, 16-byte chunk
  const uint32_t stg_w = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)&stg[wave][0]));
  const int ocol = lane / LPCO, oun = SH ? 0 : (lane % LPCO) / LPUO, och = lane % LPUO;
  const bool oun_ok = SH || sg * UPW + oun < a.n_strips;
  const int ow0 = (sg * UPW + oun) * kT;                                   // first column of the lane's unit (SH: of the pair)
  const uint32_t vw_o = static_cast<uint32_t>((oun * kT + ocol) * C + och * EPLO) * static_cast<uint32_t>(sizeof(TO));
  // add row (fp32): 4 columns per load, two loads, the second one's fourth column re-reads column 6 (SH: 8 columns of 32 channels per
  // load, the second one's six under a lane mask)
  constexpr int LPA = SH ? 8 : 16, CPA = 64 / LPA;                         // lanes per add column, add columns per load
  const int acol = lane / LPA, aun = SH ? 0 : (lane % 16) / (16 / UPW), ach = SH ? lane % 8 : lane % (16 / UPW);
  const int aun_c = (SH || sg * UPW + aun < a.n_strips) ? aun : a.n_strips - 1 - sg * UPW;
  const int acol_l = SH ? min(acol, NOUT - 1 - CPA) : min(acol, 2);
  const uint32_t vw_a = static_cast<uint32_t>((aun_c * kT + acol) * C + ach * 4) * 4u;
  const uint32_t vw_a_l = static_cast<uint32_t>((aun_c * kT + acol_l) * C + ach * 4) * 4u;
  constexpr uint64_t MASK_A_L = SH ? (1ull << ((NOUT - CPA) * LPA)) - 1ull : ~0ull;
  const float b0 = a.bias ? a.bias[c] : 0.f;

  // NR wide DMA loads of input row `hrow` of this image (0 <= hrow < H) into ring slot `slot` - always NR instructions
  auto dma_row = [&](int slot, int hrow) {
    const uint32_t dst = ring_w + static_cast<uint32_t>(slot) * SLOT;
    const long grow = n * H + hrow;
    if (grow == 0 || grow == last_row) {                                  // (wave-uniform, rare) per-lane offsets clamped into the row:
      const int row0 = static_cast<int>(img_elem) + hrow * static_cast<int>(rs);   // what a clamped lane fetches belongs to a column outside
      const int g0 = static_cast<int>(goff);                                        // the image, which the pair masks zero anyway
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int colk = (k == NR - 1) ? wcol_l + k * CPI : wcol + k * CPI;
        const int rel = min(max(g0 + (wun_c * kT + colk) * C + wch * EPL, 0), static_cast<int>(rs) - EPL);
        if (SH && k == NR - 1) dma_lds16_masked(dst + k * 1024u, static_cast<uint32_t>(row0 + rel) * static_cast<uint32_t>(sizeof(TI)), rx, 0u, MASK_IN_L);
        else dma_lds16(dst + k * 1024u, static_cast<uint32_t>(row0 + rel) * static_cast<uint32_t>(sizeof(TI)), rx, 0u);
      }
    } else {
      const uint32_t base = static_cast<uint32_t>(static_cast<long>(img_elem) + hrow * rs + goff);
      uint32_t Cs = static_cast<uint32_t>(C);
      asm volatile("" : "+s"(Cs));
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (SH && k == NR - 1) dma_lds16_masked(dst + k * 1024u, vw_in_l, rx, (base + static_cast<uint32_t>(k * CPI) * Cs) * static_cast<uint32_t>(sizeof(TI)), MASK_IN_L);
        else dma_lds16(dst + k * 1024u, k == NR - 1 ? vw_in_l : vw_in, rx, (base + static_cast<uint32_t>(k * CPI) * Cs) * static_cast<uint32_t>(sizeof(TI)));
      }
    }
  };
  // the row a slot holds -> registers (the caller has waited for its DMA)
  auto read_row = [&](uint32_t (&r)[kCols], int slot) {
    const uint32_t le = SH ? static_cast<uint32_t>(ul * kT * CSTR + (lane % CH)) : static_cast<uint32_t>(lane);   // the lane's element of column 0
    uint32_t ad = ring_w + static_cast<uint32_t>(slot) * SLOT + le * static_cast<uint32_t>(sizeof(TI));
    asm volatile("" : "+v"(ad));
    if constexpr (sizeof(TI) == 4) {
      lds_u32_t p = reinterpret_cast<lds_u32_t>(static_cast<uintptr_t>(ad));
#pragma unroll
      for (int j = 0; j < kCols; ++j) r[j] = p[j * CSTR];
    } else {
      lds_u16_t p = reinterpret_cast<lds_u16_t>(static_cast<uintptr_t>(ad));
#pragma unroll
      for (int j = 0; j < kCols; ++j) r[j] = p[j * CSTR];
    }
  };

  uint32_t win[NS][kPairs];
  // prologue: input rows r_begin - 3 .. r_begin + 2 through the ring, K at a time (full waits: six rows per band.  Measured and
  // dropped: filling the window through six compute-less steps of the pipeline below - with nothing to overlap a fill step waits
  // for its row just the same, and pays the step's bookkeeping: 143 / 165 / 203 / 150 us against 138 / 135 / 180 / 134 at 56 x 56 x 96)
#pragma unroll
  for (int j0 = 0; j0 < 6; j0 += K) {
#pragma unroll
    for (int jj = 0; jj < K; ++jj)
      if (j0 + jj < 6) dma_row(jj, min(max(r_begin - 3 + j0 + jj, 0), H - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int jj = 0; jj < K; ++jj) {
      if (j0 + jj < 6) {
        uint32_t raw[kCols];
        read_row(raw, jj);
        const int hr = r_begin - 3 + j0 + jj;
        if (hr >= 0 && hr < H) pack_row_lds(win[(j0 + jj) % NS], raw, m, TI());
        else pack_row_lds(win[(j0 + jj) % NS], raw, zero_m, TI());
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the slots are read before they are re-issued
  }
  // rows of steps 0 .. K - 1 (input rows r_begin + 3 ...) -> slots 0 .. K - 1
#pragma unroll
  for (int j = 0; j < K; ++j) dma_row(j, min(r_begin + 3 + j, H - 1));

  const long gout = static_cast<long>(sg * UPW * kT) * C + cg * CH;
  const uint32_t out_base = img_elem + static_cast<uint32_t>(gout);
  const bool ragged = W % kT != 0 || (SH && (a.n_strips & 1));           // the wavefront's columns may run beyond the row

  // the add row of image row `h` -> add slot `as` (always NA instructions)
  auto dma_add = [&](int as, int h) {
    const uint32_t dst = aring_w + static_cast<uint32_t>(as) * ASLOT;
    if (ragged && n * H + h == last_row) {                                 // (rare, wave-uniform) the last strip's columns beyond the row
      const int row0 = static_cast<int>(img_elem) + h * static_cast<int>(rs);   // would leave the tensor: per-lane offsets clamped into the row
      const int r0 = min(static_cast<int>(gout) + (aun_c * kT + acol) * C + ach * 4, static_cast<int>(rs) - 4);
      const int r1 = min(static_cast<int>(gout) + (aun_c * kT + CPA + acol_l) * C + ach * 4, static_cast<int>(rs) - 4);
      dma_lds16(dst, static_cast<uint32_t>(row0 + r0) * 4u, ra, 0u);
      if (SH) dma_lds16_masked(dst + 1024u, static_cast<uint32_t>(row0 + r1) * 4u, ra, 0u, MASK_A_L);
      else dma_lds16(dst + 1024u, static_cast<uint32_t>(row0 + r1) * 4u, ra, 0u);
    } else {
      const uint32_t so0 = (out_base + static_cast<uint32_t>(h * rs)) * 4u;
      const uint32_t so1 = (out_base + static_cast<uint32_t>(h * rs) + static_cast<uint32_t>(CPA) * static_cast<uint32_t>(C)) * 4u;
      dma_lds16(dst, vw_a, ra, so0);
      if (SH) dma_lds16_masked(dst + 1024u, vw_a_l, ra, so1, MASK_A_L);
      else dma_lds16(dst + 1024u, vw_a_l, ra, so1);
    }
  };
  if (ADD && !(DW_ABL & 2)) {
#pragma unroll
    for (int j = 0; j < AD; ++j) dma_add(j, min(r_begin + j, H - 1));      // add rows of steps 0 .. AD - 1
  }
  int sl = 0, asl = 0;                                                     // ring slot of this step's row / add row (wave-uniform)

  // counted waits of the first steps: the rows of steps 0 .. K - 1 and the add rows of steps 0 .. AD - 1 were issued back to back above
#define DWDMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define DWDMA_STEP(P)                                                                                              \
  {                                                                                                                \
    const int st = i + (P);                                               /* step number inside the band */        \
    const int h0 = r_begin + st;                                                                                   \
    const int h = min(h0, H - 1);                                                                                  \
    const bool row_ok = st < n_rows;                                      /* wave-uniform */                       \
    /* the row issued K steps ago */                                                                               \
    if ((P) < K && i == 0) DWDMA_WAIT(((P) < K ? (K - 1 - (P)) * NR : 0) + AD * NAE + (P) * PER_STEP);             \
    else DWDMA_WAIT(WAIT_ROW);                                                                                     \
    if (ADD && !(DW_ABL & 2)) {                                            /* the add row of step st + AD */       \
      int as = asl + AD;                                                                                           \
      if (as > AD) as -= AD + 1;                                                                                   \
      dma_add(as, min(h0 + AD, H - 1));                                                                            \
    }                                                                                                              \
    /* the 13 LDS reads of the entering row are issued here and consumed behind six of the seven filter rows (the row is the \
       window's LAST: only kh = 6 needs it) - their latency runs under 168 dot products instead of in front of them */          \
    uint32_t raw[kCols];                                                                                           \
    read_row(raw, sl);                                                                                             \
    float acc[kT];                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < kT; ++t) acc[t] = b0;                                                    \
    _Pragma("unroll") for (int kh = (DW_ABL & 1) ? 6 : 0; kh < 6; ++kh) {                                          \
      const uint32_t(&d)[kPairs] = win[((P) + kh) % NS];                                                           \
      _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                              \
          acc[t] = dot2(d[t / 2 + e], (t & 1) ? wo[kh][e] : we[kh][e], acc[t]);                                    \
      }                                                                                                            \
    }                                                                                                              \
    /* (the compiler moves the packing - and with it the wait for the LDS reads - in front of the dot products above unless the row's \
       registers depend on their results: an empty statement that takes both) */                                                       \
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]), \
                 "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12])                           \
                 : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]));     \
    if (h0 + 3 < H) pack_row_lds(win[((P) + 6) % NS], raw, m, TI());                                               \
    else pack_row_lds(win[((P) + 6) % NS], raw, zero_m, TI());                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     /* slot read: free for the row of step st + K */ \
    if (!(DW_ABL & 2)) dma_row(sl, min(h0 + 3 + K, H - 1));                                                        \
    {                                                                                                              \
      const uint32_t(&d)[kPairs] = win[((P) + 6) % NS];                                                            \
      _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                              \
          acc[t] = dot2(d[t / 2 + e], (t & 1) ? wo[6][e] : we[6][e], acc[t]);                                      \
      }                                                                                                            \
    }                                                                                                              \
    if (ADD) {                                                                                                     \
      if ((P) < AD && i == 0) DWDMA_WAIT(((P) < AD ? (AD - 1 - (P)) * NAE : 0) + (P) * PER_STEP + NAE + NR);       \
      else DWDMA_WAIT(WAIT_ADD);                                                                                   \
      uint32_t ad = aring_w + static_cast<uint32_t>(asl) * ASLOT +                                                 \
                    (SH ? static_cast<uint32_t>(ul * kT * 32 + (lane % CH)) : static_cast<uint32_t>(lane)) * 4u;   \
      asm volatile("" : "+v"(ad));                                                                                 \
      lds_u32_t ap = reinterpret_cast<lds_u32_t>(static_cast<uintptr_t>(ad));                                      \
      _Pragma("unroll") for (int t = 0; t < kT; ++t) acc[t] += __uint_as_float(ap[t * (SH ? 32 : 64)]);            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   /* read before the next step re-issues the slot */ \
    }                                                                                                              \
    if (DW_ABL & 4) { float z = 0.f; _Pragma("unroll") for (int t = 0; t < kT; ++t) z += acc[t]; asm volatile("" :: "v"(z)); } \
    else {                                                                /* the lane's 7 values -> stg[column][(unit, channel)] */ \
      uint32_t sa = stg_w + (SH ? static_cast<uint32_t>(ul * kT * CSTRO + (lane % CH)) : static_cast<uint32_t>(lane)) * static_cast<uint32_t>(sizeof(TO)); \
      asm volatile("" : "+v"(sa));                                                                                 \
      _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                             \
        if constexpr (sizeof(TO) == 4) reinterpret_cast<lds_w32_t>(static_cast<uintptr_t>(sa))[t * CSTRO] = __float_as_uint(acc[t]);   \
        else reinterpret_cast<lds_w16_t>(static_cast<uintptr_t>(sa))[t * CSTRO] = static_cast<uint16_t>(pack2_bf16(acc[t], 0.f));      \
      }                                                                                                            \
    }                                                                                                              \
    if (DW_ABL & 4) {                                                                                              \
    } else if (row_ok) {                                                                                           \
      uint32_t la = stg_w + static_cast<uint32_t>(lane) * 16u;                                                     \
      asm volatile("" : "+v"(la));                                                                                 \
      _Pragma("unroll") for (int k = 0; k < NSO; ++k) {                                                            \
        const u32x4_t v = *reinterpret_cast<lds_v4_t>(static_cast<uintptr_t>(la + k * 1024u));                     \
        const bool st_ok = oun_ok && k * CPIO + ocol < NOUT && ow0 + k * CPIO + ocol < W;   /* (ragged widths: the last strip's columns inside the image) */ \
        /* the compiler guards a store under EXEC with s_cbranch_execz: a store NO lane of the wavefront takes (the ragged last  \
           strip, second 16-byte group) would not be issued and the counted vmcnt waits would run two instructions short - the   \
           wave-uniform test keeps the number of VMEM instructions per step independent of EXEC */                               \
        if (__builtin_amdgcn_ballot_w64(st_ok) != 0ull) {                                                          \
          if (st_ok)                                                                                               \
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, vw_o,                                                  \
                (out_base + static_cast<uint32_t>(h * rs) + static_cast<uint32_t>(k * CPIO) * static_cast<uint32_t>(C)) * static_cast<uint32_t>(sizeof(TO)), 0); \
        } else {                                                                                                   \
          dma_lds<4>(sink_w, 0u, rx, 0u);                                                                          \
        }                                                                                                          \
      }                                                                                                            \
    } else {                                                               /* nothing to store: keep the count (loads nobody reads) */ \
      _Pragma("unroll") for (int k = 0; k < NSO; ++k) dma_lds<4>(sink_w, 0u, rx, 0u);                              \
    }                                                                                                              \
    sl = sl + 1 == K ? 0 : sl + 1;                                                                                 \
    asl = asl + 1 > AD ? 0 : asl + 1;                                                                              \
  }

  for (int i = 0; i < n_rows; i += NS) {
    DWDMA_STEP(0) DWDMA_STEP(1) DWDMA_STEP(2) DWDMA_STEP(3) DWDMA_STEP(4) DWDMA_STEP(5) DWDMA_STEP(6)
  }
#undef DWDMA_WAIT
#undef DWDMA_STEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // no DMA may land in LDS that is no longer ours
}

// prefetch depth: K = 2 rows, add row in its own step (round 4's), after measuring K = 2 .. 4 x AD = 0 .. 3 at every stage shape: no
// depth moves any of them by more than the run-to-run spread (profiles/r05_dw.md - the kernel is not waiting for latency).  Measurement
// builds: make EXTRA="-DDW_DEPTH_K=4 -DDW_DEPTH_AD=2" (three workgroups per CU need <= 53 KiB of LDS each: tools/resource_usage.py)
#ifndef DW_DEPTH_K
#define DW_DEPTH_K 2
#endif
#ifndef DW_DEPTH_AD
#define DW_DEPTH_AD 0
#endif
template <typename TI, typename TO, int CH, bool ADD, bool SH>
int launch_dma(const WinArgs& a, hipStream_t s) {
  const long blocks = a.items_per_cg / 4 * a.n_cg;
  hipLaunchKernelGGL((dwconv7x7_dma_kernel<TI, TO, CH, ADD, SH, DW_DEPTH_K, ADD ? DW_DEPTH_AD : 0>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  return static_cast<int>(hipGetLastError());
}


// ------------------------------------------------------------------------------------------------------------------------
// Filter / bias gradient in the same form:  dw[kh][kw][c] = sum_{n,h,w} dy[n,h,w,c] x[n,h+kh-3,w+kw-3,c],  db[c] = sum dy.
// lane = (strip of 7 columns, channel) keeps its 49 + 1 partial sums in registers next to the 7-row x window and walks down whole
// images (one item = (image, strip group); a wavefront takes every (4 x parts)-th item of its channel group).  Per dy row: the 7
// values of the strip are packed twice - D_q = (dy[2q], dy[2q+1]) and D'_q = (dy[2q-1], dy[2q]) - so that EVERY tap is four aligned
// dot products on the window's pairs P_i = (X[2i], X[2i+1]) (X[j] = window column j = image column w0 - 3 + j):
//     kw = 2e   : sum_t dy[t] X[t + kw] = sum_q D_q . P_{q+e}          kw = 2e + 1 : = sum_q D'_q . P_{q+e}        (q = 0..3)
// (the LDS-ring kernels rebuild three misaligned x pairs with v_alignbit per pair step instead).  196 dot products per row, as the
// forward.  At the end the two units of a CH = 32 wavefront are added (lane ^ 32), the workgroup's four wavefronts through LDS,
// and one partial per workgroup goes to ws[part][50][C]; reduce_parts_kernel (model_kernels.hip) sums the parts in a fixed order.
struct WinWgArgs {
  const void* x; const uint16_t* dy; float* ws;
  int N, H, W, C;
  int n_strips, n_sg, n_cg, parts;           // strips per row, strip groups, channel groups, workgroups (= partial sums) per channel group
  long items_per_cg;                         // N * n_sg
};

// RAGGED: W is not a multiple of 7 - the last strip's dy columns beyond the row are loaded from a clamped column and zeroed.
template <typename TX, int CH, bool RAGGED>
__global__ __launch_bounds__(256, 2) void dwconv7x7_wgrad_win_kernel(const WinWgArgs a) {
  constexpr int UPW = 64 / CH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long blk;
  {
    const long L = blockIdx.x, B = gridDim.x;
    const long q = B / 8, r = B % 8, xcd = L % 8, k = L / 8;
    blk = xcd * q + (xcd < r ? xcd : r) + k;
  }
  const int cg = static_cast<int>(blk % a.n_cg), part = static_cast<int>(blk / a.n_cg);   // channel group fastest: as the forward kernel
  const int H = a.H, W = a.W, C = a.C;
  const long rs = static_cast<long>(W) * C;
  const int ul = lane / CH;
  const uint32_t tensor_elems = static_cast<uint32_t>(static_cast<long>(a.N) * H * rs);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, tensor_elems * static_cast<uint32_t>(sizeof(TX)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.dy), 0, tensor_elems * 2u, 0x00020000);
  const long last_row = static_cast<long>(a.N) * H - 1;
  const uint32_t zero_m[kPairs] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};

  float acc[7][7], accb = 0.f;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) acc[kh][kw] = 0.f;

  for (long it = static_cast<long>(part) * 4 + wave; it < a.items_per_cg; it += static_cast<long>(a.parts) * 4) {
    const int sg = static_cast<int>(it % a.n_sg);
    const long n = it / a.n_sg;
    const bool unit_ok = sg * UPW + ul < a.n_strips;
    const int ulc = unit_ok ? ul : a.n_strips - 1 - sg * UPW;
    // pair masks (what lies outside the image is zero in the window): wave-uniform per unit, so they live in SGPRs and a CH = 32
    // wavefront selects its unit's with one v_cndmask per use - seven VGPRs less than the forward kernel's per-lane array
    uint32_t mu[UPW][kPairs];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      const int su = min(sg * UPW + u, a.n_strips - 1) * kT;
#pragma unroll
      for (int i = 0; i < kPairs; ++i) {
        const int wl = su - 3 + 2 * i, wh = wl + 1;
        mu[u][i] = ((wl >= 0 && wl < W) ? 0x0000ffffu : 0u) | ((2 * i + 1 < kCols && wh >= 0 && wh < W) ? 0xffff0000u : 0u);
      }
    }
    uint32_t m[kPairs];
#pragma unroll
    for (int i = 0; i < kPairs; ++i) m[i] = (UPW == 2 && ul) ? mu[UPW - 1][i] : mu[0][i];
    const uint32_t dmask = unit_ok ? 0xffffffffu : 0u;                   // an idle unit contributes nothing
    const uint32_t voff = static_cast<uint32_t>(ulc * kT * C + (lane % CH));
    const uint32_t vb_x = voff * static_cast<uint32_t>(sizeof(TX)), vb_d = voff * 2u;
    const long goff = static_cast<long>(sg * UPW * kT - 3) * C + cg * CH;
    const uint32_t img_elem = static_cast<uint32_t>(n * H * rs);
    const uint32_t dy_base = img_elem + static_cast<uint32_t>(static_cast<long>(sg * UPW * kT) * C + cg * CH);

    auto load_row = [&](RawRow<TX>& r, int hrow) {
      const long grow = n * H + hrow;
      if (grow == 0 || grow == last_row) {
        asm volatile("; first / last row of the tensor" ::: "memory");
        const int row0 = static_cast<int>(img_elem) + hrow * static_cast<int>(rs);
        const int rel0 = static_cast<int>(goff) + static_cast<int>(voff);
#pragma unroll
        for (int j = 0; j < kCols; ++j) {
          const int rel = min(max(rel0 + j * C, 0), static_cast<int>(rs) - 1);
          const uint32_t vo = static_cast<uint32_t>(row0 + rel) * static_cast<uint32_t>(sizeof(TX));
          if constexpr (sizeof(TX) == 4) r.v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, vo, 0, 0));
          else r.v[j] = __builtin_amdgcn_raw_buffer_load_b16(rs_x, vo, 0, 0);
        }
      } else {
        const uint32_t base = static_cast<uint32_t>(static_cast<long>(img_elem) + hrow * rs + goff);
        uint32_t Cs = static_cast<uint32_t>(C);                           // (opaque per call: the 13 column offsets are recomputed by the scalar
        asm volatile("" : "+s"(Cs));                                       //  unit instead of being kept in 13 SGPRs across the unrolled steps)
#pragma unroll
        for (int j = 0; j < kCols; ++j) {
          const uint32_t so = (base + static_cast<uint32_t>(j) * Cs) * static_cast<uint32_t>(sizeof(TX));
          if constexpr (sizeof(TX) == 4) r.v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, vb_x, so, 0));
          else r.v[j] = __builtin_amdgcn_raw_buffer_load_b16(rs_x, vb_x, so, 0);
        }
      }
    };
    const int w0r = (sg * UPW + ulc) * kT;                                // (ragged strips) first column of this lane's strip
    auto load_dy = [&](uint16_t (&d)[kT], int hrow) {
      const uint32_t base = dy_base + static_cast<uint32_t>(hrow * rs);
      uint32_t Cs = static_cast<uint32_t>(C);
      asm volatile("" : "+s"(Cs));
#pragma unroll
      for (int t = 0; t < kT; ++t) {
        if constexpr (RAGGED) {
          const uint32_t vo = vb_d + static_cast<uint32_t>(min(t, W - 1 - w0r)) * Cs * 2u;    // per-lane clamped column
          const uint16_t v = __builtin_amdgcn_raw_buffer_load_b16(rs_d, vo, base * 2u, 0);
          d[t] = w0r + t < W ? v : static_cast<uint16_t>(0);
        } else {
          d[t] = __builtin_amdgcn_raw_buffer_load_b16(rs_d, vb_d, (base + static_cast<uint32_t>(t) * Cs) * 2u, 0);
        }
      }
    };

    uint32_t win[7][kPairs];
#pragma unroll
    for (int j0 = 0; j0 < 6; j0 += 3) {
      RawRow<TX> raw[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) load_row(raw[jj], min(max(j0 + jj - 3, 0), H - 1));
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int hr = j0 + jj - 3;
        if (hr >= 0 && hr < H) pack_row(win[j0 + jj], raw[jj], m);
        else pack_row(win[j0 + jj], raw[jj], zero_m);
      }
    }
    RawRow<TX> nx;
    load_row(nx, min(3, H - 1));
    uint16_t dn[kT];
    load_dy(dn, 0);

#define DWWG_STEP(P)                                                                                               \
    {                                                                                                              \
      const int h = min(i + (P), H - 1);                                                                           \
      const bool row_ok = i + (P) < H;                                                                             \
      if (h + 3 < H) pack_row(win[((P) + 6) % 7], nx, m);                                                          \
      else pack_row(win[((P) + 6) % 7], nx, zero_m);                                                               \
      const uint32_t rm = row_ok ? dmask : 0u;                                                                     \
      uint32_t D[4], E[4];                                                                                         \
      D[0] = (static_cast<uint32_t>(dn[0]) | (static_cast<uint32_t>(dn[1]) << 16)) & rm;                           \
      D[1] = (static_cast<uint32_t>(dn[2]) | (static_cast<uint32_t>(dn[3]) << 16)) & rm;                           \
      D[2] = (static_cast<uint32_t>(dn[4]) | (static_cast<uint32_t>(dn[5]) << 16)) & rm;                           \
      D[3] = static_cast<uint32_t>(dn[6]) & rm;                                                                    \
      E[0] = (static_cast<uint32_t>(dn[0]) << 16) & rm;                                                            \
      E[1] = (static_cast<uint32_t>(dn[1]) | (static_cast<uint32_t>(dn[2]) << 16)) & rm;                           \
      E[2] = (static_cast<uint32_t>(dn[3]) | (static_cast<uint32_t>(dn[4]) << 16)) & rm;                           \
      E[3] = (static_cast<uint32_t>(dn[5]) | (static_cast<uint32_t>(dn[6]) << 16)) & rm;                           \
      load_row(nx, min(h + 4, H - 1));                                                                             \
      load_dy(dn, min(h + 1, H - 1));                                                                              \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) accb = dot2(D[q], 0x3f803f80u, accb);                          \
      _Pragma("unroll") for (int kh = 0; kh < 7; ++kh) {                                                           \
        const uint32_t(&d)[kPairs] = win[((P) + kh) % 7];                                                          \
        _Pragma("unroll") for (int kw = 0; kw < 7; ++kw) {                                                         \
          _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                            \
            acc[kh][kw] = dot2((kw & 1) ? E[q] : D[q], d[q + kw / 2], acc[kh][kw]);                                \
        }                                                                                                          \
      }                                                                                                            \
    }
    for (int i = 0; i < H; i += 7) {
      DWWG_STEP(0) DWWG_STEP(1) DWWG_STEP(2) DWWG_STEP(3) DWWG_STEP(4) DWWG_STEP(5) DWWG_STEP(6)
    }
#undef DWWG_STEP
  }

  // ---- the workgroup's partial: units of a wavefront (CH = 32), then the four wavefronts through LDS
  __shared__ float red[4][50][CH];
  if constexpr (UPW == 2) {
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) acc[kh][kw] += __shfl_xor(acc[kh][kw], 32, 64);
    accb += __shfl_xor(accb, 32, 64);
  }
  if (lane < CH) {
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) red[wave][kh * 7 + kw][lane] = acc[kh][kw];
    red[wave][49][lane] = accb;
  }
  __syncthreads();
  float* pw = a.ws + static_cast<long>(part) * 50 * C + cg * CH;
  for (int q = threadIdx.x; q < 50 * CH; q += 256) {
    const int tap = q / CH, cc = q % CH;
    pw[static_cast<long>(tap) * C + cc] = (red[0][tap][cc] + red[1][tap][cc]) + (red[2][tap][cc] + red[3][tap][cc]);
  }
}


int& win_policy() {
  static int p = getenv("APGD_DW_WIN") ? atoi(getenv("APGD_DW_WIN")) : 1;
  return p;
}

}  // namespace

int dw_shared_halo_switch(int value) {
  static int sh = !(getenv("APGD_DW_SH") && atoi(getenv("APGD_DW_SH")) == 0) ? 1 : 0;
  const int prev = sh;
  if (value >= 0) sh = value ? 1 : 0;
  return prev;
}

extern "C" int cnx_dwconv7x7_win_policy(int policy) {
  int& p = win_policy();
  const int prev = p;
  if (policy >= 0) p = policy > 0 ? 1 : 0;
  return prev;
}

// -> APGD_OK / an error of the launch; -1 when this shape is not for the window kernel (the caller goes on to the LDS kernels)
int dw_win_launch(const void* x, int x_dtype, const float* w49c, const float* bias, const float* add, void* out, int out_dtype,
                  int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, hipStream_t s) {
  // policy (cnx_dwconv7x7_win_policy / APGD_DW_WIN): 0 = never (the LDS-ring kernels of model_kernels.hip), otherwise this kernel for
  // every bf16-operand call with C % 32 == 0 and W >= 7: measured ahead of the LDS-ring kernels (and of its own first form, which
  // loaded rows into registers) at every ConvNeXt-T / -B / -L shape at 224 and 320 and at batch 128 and 256 (profiles/r04_dwwin.md)
  const int on = win_policy();
  if (!on || C % 32 != 0 || H < 1 || W < 7) return -1;
  if (x_dtype == APGD_F32 && out_dtype == APGD_F32) return -1;             // the exact-fp32 path is not a bf16 kernel's business
  if (add && x_dtype == APGD_F32) return -1;                                // (no caller: forward calls carry no add operand)
  if (static_cast<long>(N) * H * W * C >= (1L << 30)) return -1;           // 32-bit BYTE offsets into the fp32 tensors (buffer addressing)
  WinArgs a;
  a.x = x; a.w49c = w49c; a.bias = bias; a.add = add; a.out = out;
  a.N = static_cast<int>(N); a.H = H; a.W = W; a.C = C; a.flip = flip;
  // APGD_DW_SH=0 / cnx_runtime_switch(CNX_SWITCH_DW_SHARED_HALO, 0): 32-channel wavefronts (C % 64 != 0) load their two strips
  // separately, as in round 4 (measurement; default: shared halo)
  const bool sh = dw_shared_halo_switch(-1) != 0;
  const int ch = (C % 64 == 0) ? 64 : 32;
  a.n_strips = (W + kT - 1) / kT;
  a.n_sg = (a.n_strips + (64 / ch) - 1) / (64 / ch);
  a.n_cg = C / ch;
  // rows per band: as few bands as fill the chip once
  const long per_band_items = static_cast<long>(N) * a.n_cg * a.n_sg;
  int n_bands = 1;
  // (ONE round of 12 wavefronts per CU: every band pays three full memory latencies for its first six rows, and a lone round has no
  //  tail; 56 x 56 x 96, batch 256: 1 / 2 / 3 bands = 128 / 133 / 143 us, batch 128: 1 band 86, 2 bands 72)
  const long want_items = 256L * 12;
  while (n_bands < 8 && per_band_items * n_bands < want_items && (H + n_bands) / (n_bands + 1) >= 7) ++n_bands;
  const int ns = 7;
  a.band = ((H + n_bands - 1) / n_bands + ns - 1) / ns * ns;              // whole groups of 7 rows (the kernel's unrolled window rotation)
  a.n_bands = (H + a.band - 1) / a.band;
  a.items_per_cg_real = static_cast<long>(N) * a.n_sg * a.n_bands;
  a.items_per_cg = (a.items_per_cg_real + 3) / 4 * 4;
#define WIN_GO(TI, TO, ADDV) return (ch == 64) ? launch_dma<TI, TO, 64, ADDV, false>(a, s) : sh ? launch_dma<TI, TO, 32, ADDV, true>(a, s) : launch_dma<TI, TO, 32, ADDV, false>(a, s);
  if (x_dtype == APGD_F32) WIN_GO(float, uint16_t, false)
  if (out_dtype == APGD_F32) { if (add) WIN_GO(uint16_t, float, true) else WIN_GO(uint16_t, float, false) }
  if (add) WIN_GO(uint16_t, uint16_t, true)
  WIN_GO(uint16_t, uint16_t, false)
#undef WIN_GO
}

// Filter / bias gradient partials into ws[parts][50][C] (bf16 dy; x fp32 or bf16, rounded to bf16 as the convolution did).
// -> number of parts written (the caller runs reduce_parts_kernel over them), a negative value = -(1 + HIP error), or 0 when this
// shape is not for the window kernel.
int dw_win_wgrad_launch(const void* x, int x_dtype, const void* dy, float* ws, int max_parts, int64_t N, int32_t H, int32_t W, int32_t C,
                        hipStream_t s) {
  if (win_policy() == 0 || C % 32 != 0 || W < kT || H < 1) return 0;
  if (static_cast<long>(N) * H * W * C >= (1L << 30)) return 0;
  WinWgArgs a;
  a.x = x; a.dy = static_cast<const uint16_t*>(dy); a.ws = ws;
  a.N = static_cast<int>(N); a.H = H; a.W = W; a.C = C;
  const int ch = (C % 64 == 0) ? 64 : 32;
  a.n_strips = (W + kT - 1) / kT;
  a.n_sg = (a.n_strips + (64 / ch) - 1) / (64 / ch);
  a.n_cg = C / ch;
  a.items_per_cg = static_cast<long>(N) * a.n_sg;
  // three wavefronts per SIMD would be 768 workgroups of four on the chip (the kernel takes two); one partial sum per workgroup.
  // (Measured and dropped: the 7 filter rows split over three launches - 21 / 14 / 14 sums and 3- / 2-row windows per lane, 83 - 128
  //  registers, four to five wavefronts per SIMD - re-reads dy and repeats the per-row overheads three times: 281 / 125 / 74 / 52 us
  //  against 159 / 74 / 47 / 32, profiles/r04_dwwin.md.)
  long parts = (768 + a.n_cg - 1) / a.n_cg;
  if (parts > max_parts) parts = max_parts;
  if (parts * 4 > a.items_per_cg) parts = (a.items_per_cg + 3) / 4;
  a.parts = static_cast<int>(parts < 1 ? 1 : parts);
  const dim3 grid(static_cast<unsigned>(a.parts) * a.n_cg), block(256);
#define WGK(TXX, CHH) { if (W % kT == 0) hipLaunchKernelGGL((dwconv7x7_wgrad_win_kernel<TXX, CHH, false>), grid, block, 0, s, a); \
                        else hipLaunchKernelGGL((dwconv7x7_wgrad_win_kernel<TXX, CHH, true>), grid, block, 0, s, a); }
  if (x_dtype == APGD_F32) { if (ch == 64) WGK(float, 64) else WGK(float, 32) }
  else { if (ch == 64) WGK(uint16_t, 64) else WGK(uint16_t, 32) }
#undef WGK
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? a.parts : -(1 + static_cast<int>(e));
}
