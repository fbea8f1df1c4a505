#!/usr/bin/env python3
"""Depthwise-7x7 kernels at the ConvNeXt-T stage shapes (batch 256): forward, input gradient (+residual add), filter gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
S = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3


code = {torch.float32: 0, torch.bfloat16: 1}
SHAPES = ((96, 56), (192, 28), (384, 14), (768, 7))
if len(sys.argv) > 2:                                   # e.g. "768x20,1536x10,192x80,384x40" (ConvNeXt-L @320)
    SHAPES = tuple(tuple(int(v) for v in t.split("x")) for t in sys.argv[2].split(","))
for C, HW in SHAPES:
    n = B * HW * HW * C
    w = torch.randn(49, C, device="cuda")
    b = torch.randn(C, device="cuda")
    xs = {torch.float32: torch.randn(B, HW, HW, C, device="cuda"), torch.bfloat16: torch.randn(B, HW, HW, C, device="cuda").bfloat16()}
    add = torch.randn(B, HW, HW, C, device="cuda")
    line = [f"C={C:4d} HW={HW:2d}"]
    for xi, xo, use_add, flip, tag in ((torch.float32, torch.bfloat16, False, 0, "fwd f32->bf16"), (torch.bfloat16, torch.bfloat16, False, 0, "fwd bf16->bf16"),
                                       (torch.bfloat16, torch.float32, True, 1, "dgrad bf16->f32+add"), (torch.bfloat16, torch.bfloat16, False, 1, "dgrad bf16->bf16")):
        out = torch.empty(B, HW, HW, C, device="cuda", dtype=xo)
        x = xs[xi]
        fn = lambda: lib.cnx_dwconv7x7_nhwc(x.data_ptr(), code[xi], w.data_ptr(), b.data_ptr(), add.data_ptr() if use_add else None,
                                            out.data_ptr(), code[xo], B, HW, HW, C, flip, S())
        t = timeit(fn)
        byts = n * (x.element_size() + out.element_size() + (4 if use_add else 0))
        line.append(f"{tag} {t:7.1f} us ({byts / t / 1e3:5.0f} GB/s)")
    ws = torch.empty(lib.cnx_dwconv7x7_wgrad_ws_floats(C), device="cuda")
    g49, db = torch.empty(49, C, device="cuda"), torch.empty(C, device="cuda")
    for xi in (torch.float32, torch.bfloat16):
        x, dy = xs[xi], xs[torch.bfloat16]
        t = timeit(lambda: lib.cnx_dwconv7x7_wgrad_nhwc(x.data_ptr(), code[xi], dy.data_ptr(), 1, g49.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                        B, HW, HW, C, S()))
        line.append(f"wgrad x={'f32' if xi == torch.float32 else 'bf16'} {t:7.1f} us")
    print(" | ".join(line), flush=True)
