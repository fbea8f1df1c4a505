#!/usr/bin/env python3
"""First stem convolution (3->48, 3x3 s2) at batch 256, 224x224: hand-written image kernels vs the library path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import revisiting_at_amd as R
from revisiting_at_amd import ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3


B = 256
x = torch.rand(B, 3, 224, 224, device="cuda").requires_grad_()
conv = torch.nn.Conv2d(3, 48, 3, 2, 1).cuda().to(memory_format=torch.channels_last)
with torch.autocast("cuda", dtype=torch.bfloat16):
    def ours_f():
        with torch.no_grad():
            return ops.stem_conv(x, conv.weight, conv.bias)
    def lib_f():
        torch.clear_autocast_cache()
        with torch.no_grad():
            return conv(x)
    def ours_b():
        with ops.input_grad_only():
            o = ops.stem_conv(x, conv.weight, conv.bias)
            torch.autograd.grad(o, x, torch.ones_like(o))
    def lib_b():
        torch.clear_autocast_cache()
        o = conv(x)
        torch.autograd.grad(o, x, torch.ones_like(o))
    print(f"forward: ours {timeit(ours_f):.1f} us   library {timeit(lib_f):.1f} us")
    print(f"forward + input gradient: ours {timeit(ours_b):.1f} us   library {timeit(lib_b):.1f} us")
