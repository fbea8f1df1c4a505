#!/usr/bin/env python3
"""First ConvStem convolution (3 -> 48, 3x3 / 2) at batch 256, 224 x 224: the convolution alone, followed by the LayerNorm + GELU kernel,
and the one-kernel form (cnx_stem_conv_ln_gelu_fwd); gradient-free calls, HIP events, median of 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand(B, 3, 224, 224, device="cuda")
w = torch.randn(48, 3, 3, 3, device="cuda") * 0.3
b = torch.randn(48, device="cuda") * 0.1
lw = torch.ones(48, device="cuda"); lb = torch.zeros(48, device="cuda")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3


with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    t_conv = timeit(lambda: R.ops.stem_conv(x, w, b))
    t_two = timeit(lambda: R.ops.layer_norm_cf_gelu(R.ops.stem_conv(x, w, b), lw, lb, 1e-6))
    t_one = timeit(lambda: R.ops.stem_conv_ln_gelu(x, w, b, lw, lb, 1e-6))
xg = x.clone().requires_grad_()
with torch.autocast("cuda", dtype=torch.bfloat16):
    t_one_g = timeit(lambda: R.ops.stem_conv_ln_gelu(xg, w, b, lw, lb, 1e-6))
    t_two_g = timeit(lambda: R.ops.layer_norm_cf_gelu(R.ops.stem_conv(xg, w, b), lw, lb, 1e-6))
print(f"B={B}: conv {t_conv:.1f} us | conv + LN/GELU kernel {t_two:.1f} us | one kernel {t_one:.1f} us | with saved tensors: two {t_two_g:.1f}, one {t_one_g:.1f}")

# filter / bias gradient: cnx_stem_conv_wgrad (round 5) against the library's convolution_backward on the same operands
lib = R._lib.load()
dy = torch.randn(B, 112, 112, 48, device="cuda").to(torch.bfloat16)
dw = torch.empty(48, 3, 3, 3, device="cuda"); db = torch.empty(48, device="cuda")
ws = torch.empty(lib.cnx_stem_conv_wgrad_ws_floats(48), device="cuda")
S = torch.cuda.current_stream().cuda_stream
t_wg = timeit(lambda: lib.cnx_stem_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 224, 224, 48, S))


def lib_wgrad():
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), xb, w.to(torch.bfloat16), [48], [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, True])


t_lib = timeit(lib_wgrad)
print(f"B={B}: filter gradient: cnx_stem_conv_wgrad {t_wg:.1f} us (462 MB: {0.462e3 * B / 256 / t_wg:.2f} TB/s) | library (cast + layout copy + MIOpen) {t_lib:.1f} us")
