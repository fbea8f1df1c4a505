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
