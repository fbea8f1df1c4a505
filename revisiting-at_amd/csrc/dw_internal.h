// dw_internal.h — internal (non-ABI) entry of the register-window depthwise kernels (dwwin_kernels.hip), called by the C ABI
// function cnx_dwconv7x7_nhwc (model_kernels.hip) before it falls back to the LDS-ring kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// -> APGD_OK (0) / a HIP launch error; -1 when the shape / dtype combination is not for the window kernel
int dw_win_launch(const void* x, int x_dtype, const float* w49c, const float* bias, const float* add, void* out, int out_dtype,
                  int64_t N, int32_t H, int32_t W, int32_t C, int32_t flip, hipStream_t s);
