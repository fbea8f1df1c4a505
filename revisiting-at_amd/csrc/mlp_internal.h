// mlp_internal.h — shared between block_kernels.hip (C ABI, first-generation kernels) and mlp_kernels.hip
// (persistent second-generation kernels).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
  long M;
  int dbg;                 // timing experiments only (APGD_BLK_DBG)
};

// barrier-free persistent forward with the weights resident in LDS (C = 96).  Returns a launch status (0 = ok), -100 if C
// is not covered.
int mlp2_fwd_launch(const BlkFwdArgs& a, int C, int resid_dtype, int out_dtype, hipStream_t s);
