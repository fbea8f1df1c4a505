// dw_internal.h — internal (non-ABI) entry of the register-window depthwise kernels (dwwin_kernels.hip), called by the C ABI
// function cnx_dwconv7x7_nhwc (model_kernels.hip) before it falls back to the LDS-ring kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// -> APGD_OK (0) / a HIP launch error; -1 when the shape / dtype combination is not for the window kernel
int dw_win_launch(const void* x, int x_dtype, const float* w49c, const float* bias, const float* add, void* out, int out_dtype,
                  int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, hipStream_t s);

// Filter / bias gradient partials ws[parts][50][C] by the register-window kernel: -> parts (> 0), 0 = shape not supported, < 0 = -(1 + HIP error)
int dw_win_wgrad_launch(const void* x, int x_dtype, const void* dy, float* ws, int max_parts, int64_t N, int32_t H, int32_t W, int32_t C,
                        hipStream_t s);

// Shared column halo of the 32-channel wavefronts (round 5) on / off; -> the previous setting, a negative value only queries
// (behind cnx_runtime_switch(CNX_SWITCH_DW_SHARED_HALO, .), block_kernels.hip).
int dw_shared_halo_switch(int value);
