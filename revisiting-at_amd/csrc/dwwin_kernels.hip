// dwwin_kernels.hip — depthwise 7x7 on bf16 operands with a REGISTER sliding window (round 4; gfx950 / MI355X).
// Reference arithmetic: /root/reference/models/convnext.py:28, 39 (`Conv2d(dim, dim, 7, padding=3, groups=dim)` under autocast:
// bf16 inputs and weights, fp32 accumulation), forward and - with the 180-degree rotated filter - its input gradient, with the
// residual gradient added in the same pass.
//
// Why another form.  The LDS-ring kernels of model_kernels.hip (dwconv7x7_roll / _multi) stage every input row through
// registers into a pair-packed LDS ring and back, in workgroup-wide phases separated by barriers; with their prefetch registers
// they sit at 220 - 256 VGPRs (+ up to 66 AGPRs): ONE or TWO wavefronts per SIMD, 4 - 8 per CU.  Round 3 measured them at half
// of both of their bounds (56x56x96 forward 145 us against 73 us of HBM time and ~60 us of v_dot2 time): with so few wavefronts
// nothing covers a phase's memory latency, and a CU never has the ~50 KB in flight that its share of the HBM bandwidth needs.
//
// Here a wavefront is on its own - no LDS, no barrier:
//   * lane = (unit, channel): CH = 32 or 64 channels of one STRIP of 7 output columns (56, 28, 14 and 7 - every ConvNeXt map at 224 -
//     are multiples of 7: no ragged strip, no idle lane; "unit" = strip of one image, a wavefront holds 64 / CH units).  NHWC
//     memory gives every load / store instruction one full 128-byte (fp32) or 64-byte (bf16) run per unit and position.
//   * the lane keeps a 7-row x 13-column window of ITS channel in 49 registers, packed as pairs of W-adjacent bf16 per dword
//     (columns 7 s - 3 ... 7 s + 9 of strip s), walks DOWN a band of the image, and per output row issues 13 loads for the row
//     that enters the window (one row ahead of its use), 7 x 28 v_dot2_f32_bf16 on the window, and the 7 stores of the finished row:
//         even t = 2 q  : pairs q .. q+3 . {(f0,f1),(f2,f3),(f4,f5),(f6,0)}
//         odd  t = 2 q+1: pairs q .. q+3 . {(0,f0),(f1,f2),(f3,f4),(f5,f6)}            (as the LDS kernels; no realignment)
//   * ~150 registers: three wavefronts per SIMD, twelve per CU, each with its own loads in flight.
// The price is the column halo (13 loads per 7 outputs: neighbouring strips sit in the same wavefront / workgroup, the re-reads
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

__device__ __forceinline__ void store_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_out(uint16_t* p, float v) { *p = static_cast<uint16_t>(pack2_bf16(v, 0.f)); }

struct WinArgs {
  const void* x; const float* w49c; const float* bias; const float* add; void* out;
  int N, H, W, C, flip;
  int n_strips, n_sg, n_cg, band, n_bands;   // strips per image row, strip groups (of 64 / CH strips), channel groups, rows per band, bands
  long items_per_cg, items_per_cg_real;      // wavefront work items of one channel group: N * n_bands * n_sg, padded to a multiple of 4
};

// One wavefront = one item: (image, band, channel group, strip group), strip group fastest.
// RAGGED: W is not a multiple of 7 (the last strip's stores are predicated per column, its group may hold an idle unit).
// Three wavefronts per SIMD (137 - 154 registers) except for the add variants with ragged strips or 32-channel groups (~180 registers, two:
// capped at 168 the CH = 32 add variant measured 203 - 209 us against 189 at 56 x 56 x 96).
template <typename TI, typename TO, int CH, bool ADD, bool RAGGED, int R>
__global__ __launch_bounds__(256, ((ADD && (RAGGED || CH == 32)) || R == 2) ? 2 : 3)
void dwconv7x7_win_kernel(const WinArgs a) {
  constexpr int UPW = 64 / CH;
  const int lane = threadIdx.x & 63;
  // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch); XCD k takes the k-th contiguous eighth of the item list,
  // so that the bands / strips of an image - which share halo rows and columns - meet in one L2
  long blk;
  {
    const long L = blockIdx.x, B = gridDim.x;
    const long q = B / 8, r = B % 8, xcd = L % 8, k = L / 8;
    blk = xcd * q + (xcd < r ? xcd : r) + k;
  }
  // (everything derived from the item is wave-uniform: say so, and the addresses below become SGPR base + one VGPR offset)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // four wavefronts = four items of ONE channel group, whose packed filter they share; item order strip group fastest, then band,
  // image, channel group slowest (the launcher pads a group's items to a multiple of four).  An XCD's contiguous eighth of the
  // workgroup list keeps the strips / bands of an image in one L2.  (Measured and dropped: a workgroup = the three channel groups
  // of one strip group at C = 96, so that a position's 384 bytes are read together - 3 - 15 % slower, profiles/r04_dwwin.md.)
  const long item = blk * 4 + wave;
  const int cg = static_cast<int>(item / a.items_per_cg);
  const long it = item % a.items_per_cg;
  const bool item_ok = it < a.items_per_cg_real;
  const int sg = static_cast<int>(it % a.n_sg);
  const int bd = static_cast<int>((it / a.n_sg) % a.n_bands);
  const long n = item_ok ? it / (static_cast<long>(a.n_sg) * a.n_bands) : 0;
  const int H = a.H, W = a.W, C = a.C;
  const int ul = lane / CH;                                                // unit of this lane inside the wavefront
  const bool unit_ok = sg * UPW + ul < a.n_strips;                         // (an odd strip count leaves the last CH = 32 wavefront half idle)
  const int ulc = unit_ok ? ul : a.n_strips - 1 - sg * UPW;               // an idle unit shadows the last strip (loads stay in range)
  const int w0 = (sg * UPW + ulc) * kT;
  const int c = cg * CH + (lane % CH);
  const int r_begin = bd * a.band, r_end = min(H, r_begin + a.band);
  const int n_rows = r_end - r_begin;
  const long rs = static_cast<long>(W) * C;                               // row stride in elements

  // ---- pair masks: window column j (0..12) is image column w0 - 3 + j; what lies outside the image is zero in the window,
  //      whatever was loaded for it (the loads of such a column fetch a neighbouring row's element: valid memory, except in the
  //      first / last row of the whole tensor, which take the clamped loads below)
  uint32_t m[kPairs];
#pragma unroll
  for (int i = 0; i < kPairs; ++i) {
    const int wl = w0 - 3 + 2 * i, wh = wl + 1;
    m[i] = ((wl >= 0 && wl < W) ? 0x0000ffffu : 0u) | ((2 * i + 1 < kCols && wh >= 0 && wh < W) ? 0xffff0000u : 0u);
  }
  const uint32_t zero_m[kPairs] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};

  // addresses = wave-uniform pointer (image, row, first column of the strip group, channel group, column j: scalar arithmetic)
  //           + ONE per-lane element offset (unit inside the group, channel inside the group), the same for every load of the lane
  const uint32_t voff = static_cast<uint32_t>(ulc * kT * C + (lane % CH));   // elements
  // (byte offsets: `uniform pointer + zero-extended 32-bit VGPR` is what selects the SGPR-base addressing mode - one VGPR, no
  //  per-load 64-bit address arithmetic; an ELEMENT index would be scaled after the extension and lose the form)
  const uint32_t vb_in = voff * static_cast<uint32_t>(sizeof(TI)), vb_out = voff * static_cast<uint32_t>(sizeof(TO)), vb_add = voff * 4u;
  const long goff = static_cast<long>(sg * UPW * kT - 3) * C + cg * CH;   // window column 0 of the group's first strip (may be < 0)
  const long last_row = static_cast<long>(a.N) * H - 1;

  // Buffer addressing (raw buffer resources over the whole tensors): address = resource base + one per-lane 32-bit byte offset
  // (VGPR) + a wave-uniform byte offset (SGPR, scalar arithmetic) - no per-load 64-bit vector address (the flat form cost one
  // v_lshl_add_u64 and a register pair per load / store: 27 of ~280 VALU instructions per output row).  Nothing relies on the
  // hardware's range check: every offset handed over is inside the tensor (the rows that could leave it take the clamped path).
  const uint32_t tensor_elems = static_cast<uint32_t>(static_cast<long>(a.N) * H * rs);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, tensor_elems * static_cast<uint32_t>(sizeof(TI)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, tensor_elems * static_cast<uint32_t>(sizeof(TO)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.add), 0, ADD ? tensor_elems * 4u : 0u, 0x00020000);
  const uint32_t img_elem = static_cast<uint32_t>(n * H * rs);             // first element of this image
  auto load_row = [&](RawRow<TI>& r, int hrow) {                          // hrow: a row of this image, 0 <= hrow < H
    const long grow = n * H + hrow;
    if (grow == 0 || grow == last_row) {                                  // (wave-uniform, rare) element index clamped into the row:
      asm volatile("; first / last row of the tensor" ::: "memory");      // nothing outside the tensor is read.  (The asm keeps this a
      const int row0 = static_cast<int>(img_elem) + hrow * static_cast<int>(rs);   // branch; 32-bit per-lane offsets, no 64-bit pointers:
      const int rel0 = static_cast<int>(goff) + static_cast<int>(voff);            // the rare path must not set the kernel's register count)
#pragma unroll
      for (int j = 0; j < kCols; ++j) {
        const int rel = min(max(rel0 + j * C, 0), static_cast<int>(rs) - 1);
        const uint32_t vo = static_cast<uint32_t>(row0 + rel) * static_cast<uint32_t>(sizeof(TI));
        if constexpr (sizeof(TI) == 4) r.v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, vo, 0, 0));
        else r.v[j] = __builtin_amdgcn_raw_buffer_load_b16(rs_x, vo, 0, 0);
      }
    } else {
      const uint32_t base = static_cast<uint32_t>(static_cast<long>(img_elem) + hrow * rs + goff);   // >= 0: not the tensor's first row
      uint32_t Cs = static_cast<uint32_t>(C);                             // (opaque per call: the column offsets are recomputed by the scalar unit
      asm volatile("" : "+s"(Cs));                                         //  instead of living in 13 SGPRs across the unrolled steps)
#pragma unroll
      for (int j = 0; j < kCols; ++j) {
        const uint32_t so = (base + static_cast<uint32_t>(j) * Cs) * static_cast<uint32_t>(sizeof(TI));
        if constexpr (sizeof(TI) == 4) r.v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, vb_in, so, 0));
        else r.v[j] = __builtin_amdgcn_raw_buffer_load_b16(rs_x, vb_in, so, 0);
      }
    }
  };

  // ---- packed filter of the workgroup's channel group in LDS: wl[kh][parity][channel] = 4 dwords (16 bytes, one ds_read_b128 per
  //      lane, conflict-free; the two units of a CH = 32 wavefront read the same addresses).  56 dwords per channel would be a
  //      third of the register budget of three wavefronts per SIMD; here a filter row lives in registers only while it is used.
  __shared__ uint4 wl[7 * 2 * CH];
  for (int q = threadIdx.x; q < 7 * CH; q += 256) {
    const int kh = q / CH, cc = q % CH;
    float f[7];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int tap = kh * 7 + kw;
      f[kw] = a.w49c[(a.flip ? 48 - tap : tap) * C + cg * CH + cc];
    }
    wl[(kh * 2 + 0) * CH + cc] = make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], 0.f));
    wl[(kh * 2 + 1) * CH + cc] = make_uint4(pack2_bf16(0.f, f[0]), pack2_bf16(f[1], f[2]), pack2_bf16(f[3], f[4]), pack2_bf16(f[5], f[6]));
  }
  __syncthreads();
  if (!item_ok) return;                                                   // (padding wavefront of the channel group's last workgroup)
  // LDS byte address of this lane's channel; laundered through an empty asm in every step so that the filter reads are not
  // loop-invariant to the compiler (it would hoist all 14 of them out of the row loop: 56 registers again)
  typedef const __attribute__((address_space(3))) u32x4_t* lds_u4_t;
  const uint32_t wl_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)wl)) +
                           (lane % CH) * 16u;
  const float b0 = a.bias ? a.bias[c] : 0.f;

  // ---- window: NS = 6 + R slots; slot k holds input row r_begin - 3 + j for j % NS == k.  A step produces R output rows: with
  //      R = 2 a wavefront has TWO rows of loads in flight (the latency of a row's loads is what a step waits for: profiles/
  //      r04_dwwin.md) and every filter row fetched from LDS serves both output rows.
  constexpr int NS = 6 + R;
  uint32_t win[NS][kPairs];
  // prologue: input rows j = 0 .. 5 (rows r_begin - 3 .. r_begin + 2), three at a time; rows above / below the image are zeros
#pragma unroll
  for (int j0 = 0; j0 < 6; j0 += 3) {
    RawRow<TI> raw[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) load_row(raw[jj], min(max(r_begin - 3 + j0 + jj, 0), H - 1));
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int hr = r_begin - 3 + j0 + jj;
      if (hr >= 0 && hr < H) pack_row(win[j0 + jj], raw[jj], m);          // wave-uniform
      else pack_row(win[j0 + jj], raw[jj], zero_m);
    }
  }
  RawRow<TI> nx[R];                                                       // the rows that enter the window next (input rows j = i + 6 ...)
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(nx[r], min(r_begin + 3 + r, H - 1));

  const long gout = static_cast<long>(sg * UPW * kT) * C + cg * CH;       // first output column of the group's first strip
  TO* oimg = static_cast<TO*>(a.out) + n * H * rs + gout;
  const float* aimg = ADD ? a.add + n * H * rs + gout : nullptr;
  const uint32_t out_base = img_elem + static_cast<uint32_t>(gout);       // element index of the group's first output column, row 0

  // ---- one step = R output rows.  P = i % NS (compile time): window slots are register arrays, their indices must be static.
#define DWWIN_STEP(P)                                                                                              \
  {                                                                                                                \
    const int h0 = r_begin + i + (P);                                     /* first output row of the step */        \
    /* the rows prefetched one step ago (input rows h0 + 3 ...) enter slots (P + 6 ...) % NS */                     \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                \
      if (h0 + 3 + r < H) pack_row(win[((P) + 6 + r) % NS], nx[r], m);                                             \
      else pack_row(win[((P) + 6 + r) % NS], nx[r], zero_m);                                                       \
    }                                                                                                              \
    /* prefetch the input rows of the next step */                                                                 \
    _Pragma("unroll") for (int r = 0; r < R; ++r) load_row(nx[r], min(h0 + 3 + R + r, H - 1));                     \
    float acc[R][kT];                                                                                              \
    _Pragma("unroll") for (int r = 0; r < R; ++r)                                                                  \
      _Pragma("unroll") for (int t = 0; t < kT; ++t) acc[r][t] = b0;                                               \
    uint32_t wa = wl_addr;                                                                                         \
    asm volatile("" : "+v"(wa));                                                                                   \
    lds_u4_t wlane = reinterpret_cast<lds_u4_t>(static_cast<uintptr_t>(wa));                                       \
    /* window slot q of the step holds input row h0 - 3 + q: filter row kh = q - r of output row h0 + r */         \
    _Pragma("unroll") for (int kh = 0; kh < 7; ++kh) {                                                             \
      const u32x4_t we4 = wlane[(kh * 2 + 0) * CH], wo4 = wlane[(kh * 2 + 1) * CH];                                \
      const uint32_t we[4] = {we4.x, we4.y, we4.z, we4.w}, wo[4] = {wo4.x, wo4.y, wo4.z, wo4.w};                   \
      _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                              \
        const uint32_t(&d)[kPairs] = win[((P) + kh + r) % NS];                                                     \
        _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                           \
          _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                            \
            acc[r][t] = dot2(d[t / 2 + e], (t & 1) ? wo[e] : we[e], acc[r][t]);                                    \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                \
      const int h = min(h0 + r, H - 1);                                                                            \
      const bool row_ok = i + (P) + r < n_rows;                           /* wave-uniform */                       \
      float av[kT];                                                                                                \
      if (ADD) {              /* after the arithmetic (compiler barrier): 7 registers less across the dot products */ \
        asm volatile("" ::: "memory");                                                                             \
        const float* ap = aimg + h * rs;                                                                           \
        _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                           \
          if (RAGGED) av[t] = ap[voff + static_cast<long>(min(t, W - 1 - w0)) * C];                                \
          else av[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, vb_add,                                  \
                           (out_base + static_cast<uint32_t>(h * rs) + static_cast<uint32_t>(t) * static_cast<uint32_t>(C)) * 4u, 0)); \
        }                                                                                                          \
      }                                                                                                            \
      if (unit_ok && row_ok) {                                                                                     \
        TO* op = oimg + h * rs;                                                                                    \
        _Pragma("unroll") for (int t = 0; t < kT; ++t) {                                                           \
          const float ov = ADD ? acc[r][t] + av[t] : acc[r][t];                                                    \
          if (RAGGED) {                                                                                            \
            if (w0 + t < W) store_out(reinterpret_cast<TO*>(reinterpret_cast<char*>(op + static_cast<long>(t) * C) + vb_out), ov); \
          } else {                                                                                                 \
            const uint32_t so = (out_base + static_cast<uint32_t>(h * rs) + static_cast<uint32_t>(t) * static_cast<uint32_t>(C)) * \
                                static_cast<uint32_t>(sizeof(TO));                                                 \
            if constexpr (sizeof(TO) == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ov), rs_o, vb_out, so, 0);  \
            else __builtin_amdgcn_raw_buffer_store_b16(static_cast<uint16_t>(pack2_bf16(ov, 0.f)), rs_o, vb_out, so, 0);     \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }

  // (bands are multiples of NS rows except possibly the last of an image: its surplus steps compute on clamped rows and store nothing)
  for (int i = 0; i < n_rows; i += NS) {
    if constexpr (R == 1) {
      DWWIN_STEP(0) DWWIN_STEP(1) DWWIN_STEP(2) DWWIN_STEP(3) DWWIN_STEP(4) DWWIN_STEP(5) DWWIN_STEP(6)
    } else {
      DWWIN_STEP(0) DWWIN_STEP(2) DWWIN_STEP(4) DWWIN_STEP(6)
    }
  }
#undef DWWIN_STEP
}

template <typename TI, typename TO, int CH, bool ADD>
int launch_win(const WinArgs& a, int rows_per_step, hipStream_t s) {
  const long blocks = a.items_per_cg / 4 * a.n_cg;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define WIN_K(RG, RR) hipLaunchKernelGGL((dwconv7x7_win_kernel<TI, TO, CH, ADD, RG, RR>), grid, block, 0, s, a)
  // (R = 2 - two output rows per step, two rows of loads in flight per wavefront - measured on MI355X: needs ~180 registers, i.e.
  //  two wavefronts per SIMD, and is 10 - 40 % SLOWER than R = 1 at three (profiles/r04_dwwin.md): only R = 1 is instantiated)
  (void)rows_per_step;
  if (a.W % kT == 0) WIN_K(false, 1);
  else WIN_K(true, 1);
#undef WIN_K
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
  const int cg = static_cast<int>(blk / a.parts), part = static_cast<int>(blk % a.parts);
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

extern "C" int cnx_dwconv7x7_win_policy(int policy) {
  int& p = win_policy();
  const int prev = p;
  if (policy >= 0) p = policy > 2 ? 2 : policy;
  return prev;
}

// -> APGD_OK / an error of the launch; -1 when this shape is not for the window kernel (the caller goes on to the LDS kernels)
int dw_win_launch(const void* x, int x_dtype, const float* w49c, const float* bias, const float* add, void* out, int out_dtype,
                  int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, hipStream_t s) {
  // policy (cnx_dwconv7x7_win_policy / APGD_DW_WIN): 0 = never, 1 = where it measured ahead of the LDS-ring kernels (default), 2 = every
  // shape it supports
  const int on = win_policy();
  if (!on || C % 32 != 0 || H < 1 || W < 7) return -1;
  if (on == 1) {
    // Measured on MI355X (profiles/r04_dwwin.md; us, window kernel vs LDS-ring kernels): ahead at every map up to 40 x 40 -
    // 28x28x192: 72 vs 113 (fwd), 96 vs 136 (dgrad + add);  14x14x384: 44 vs 49, 48 vs 63;  7x7x768: 24 vs 35, 26 vs 45;  ConvNeXt-B / -L
    // shapes alike (28x28x256 108 vs 148, 40x40x384 168 vs 245, 10x10x1536 52 vs 71) - and at 56 x 56 for the batch-128 chunks of the
    // two-stream attack (72 vs 83, 95 vs 101).  NOT ahead: 56 x 56 at batch 256 (x96: 166 vs 156, 189 vs 187; x128: 218 vs 208) and
    // the ragged-width input gradient + add above 10 x 10 (its ~180 registers leave two wavefronts per SIMD: 80x80x192 514 vs 465,
    // 20x20x768 170 vs 119).
    const long px = static_cast<long>(H) * W;
    if (px >= 2000 && px < 4000 && N > 128) return -1;
    if (add && W % kT != 0 && px >= 150) return -1;
  }
  if (x_dtype == APGD_F32 && out_dtype == APGD_F32) return -1;             // the exact-fp32 path is not a bf16 kernel's business
  if (add && x_dtype == APGD_F32) return -1;                                // (no caller: forward calls carry no add operand)
  if (static_cast<long>(N) * H * W * C >= (1L << 30)) return -1;           // 32-bit BYTE offsets into the fp32 tensors (buffer addressing)
  WinArgs a;
  a.x = x; a.w49c = w49c; a.bias = bias; a.add = add; a.out = out;
  a.N = static_cast<int>(N); a.H = H; a.W = W; a.C = C; a.flip = flip;
  const int ch = (C % 64 == 0) ? 64 : 32;
  a.n_strips = (W + kT - 1) / kT;
  a.n_sg = (a.n_strips + (64 / ch) - 1) / (64 / ch);
  a.n_cg = C / ch;
  // rows per band: whole image for the small maps; for the large ones as few bands as give >= ~3 rounds of 12 wavefronts per CU
  // (every band re-reads 6 halo rows)
  constexpr int band_env = 0;
  const long per_band_items = static_cast<long>(N) * a.n_cg * a.n_sg;
  int n_bands = 1;
  while (n_bands < 8 && per_band_items * n_bands < 3L * 256 * 12 && (H + n_bands) / (n_bands + 1) >= 7) ++n_bands;
  const int rps = 1;                                                       // output rows per step
  const int ns = 6 + rps;
  a.band = band_env > 0 ? band_env : ((H + n_bands - 1) / n_bands + ns - 1) / ns * ns;   // whole groups of NS rows (the kernel's unrolled window rotation)
  a.n_bands = (H + a.band - 1) / a.band;
  a.items_per_cg_real = static_cast<long>(N) * a.n_sg * a.n_bands;
  a.items_per_cg = (a.items_per_cg_real + 3) / 4 * 4;
#define WIN_GO(TI, TO, ADDV) return (ch == 64) ? launch_win<TI, TO, 64, ADDV>(a, rps, s) : launch_win<TI, TO, 32, ADDV>(a, rps, s);
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
