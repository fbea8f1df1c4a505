// mlp_internal.h — argument block and layout switches of the fused LN + MLP kernels (block_kernels.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 1: the forward's hidden loop is software-pipelined and the packed weights are laid out for it - slice t = [W1(t) | W2(t-1)],
// t = 0..C/8, the two pieces that do not exist zero-filled (block_kernels.hip, pack_fwd_kernel).  0: slice s = [W1(s) | W2(s)].
#ifndef BLK_FWD_PIPE
#define BLK_FWD_PIPE 1
#endif
// Widths that use it: measured on one box (profiles/r02_power_and_overlap.md) the pipelined loop wins from C = 128 up (4 % ... 20 %
// at C = 384) and loses 3 - 8 % at C = 96, where the kernel sits at the package power cap and the unpacked GELU's extra instructions
// cost more than the overlap returns.
__host__ __device__ constexpr bool blk_fwd_pipe(int C) { return BLK_FWD_PIPE != 0 && C > 96; }

struct BlkFwdArgs {
  const uint16_t* u;       // [M, C] bf16: depthwise-conv output (ln_w != NULL) or already-normalised rows
  const float* ln_w;       // [C] or NULL
  const float* ln_b;       // [C]
  float eps;
  float* mean;             // [M] or NULL (written when LN is applied)
  float* rstd;             // [M] or NULL
  const uint16_t* Wf;      // packed weights
  const float* b1;         // [4C]
  const float* b2;         // [C]
  const float* gamma;      // [C] or NULL
  const void* resid;       // [M, C] TX or NULL
  void* out;               // [M, C] TO
  uint16_t* y2;            // [M, C] bf16 pre-gamma fc2 output, or NULL
  uint16_t* hpre;          // pipelined kernels only: workspace for Hpre = LN(u) W1^T + b1 in accumulator order (see cnx_block_mlp_fwd_hpre), or NULL
  uint16_t* hact;          // training forward (WS == 2): H = GELU(Hpre) in the same tiles, for the weight-gradient contraction dW2 = dO^T H
  uint16_t* a_out;         // training forward (WS == 2): LN(u) rows [M, C] bf16, for dW1 = dHpre^T LN(u)
  long M;
  int dbg;                 // timing experiments only (APGD_BLK_DBG)
};


// widths served by the wavefront-pair Hpre backward (block_bwd_kernels.hip): -> the previous mask, a negative value only queries
int blk2b_widths_switch(int value);
int tn_pair_ring_switch(int value);   // wgrad_kernels.hip: CNX_SWITCH_TN_PAIR_RING
int gemm_nt_tile_switch(int value);   // gemm_kernels.hip: CNX_SWITCH_GEMM_NT_TILE
