#!/usr/bin/env python3
"""LayerNorm kernels of the ConvStem / downsample layers at their batch-256 shapes: forward and backward (with parameter gradients),
HIP events, median of 10; bytes = what each call has to read and write once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
S = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
code = {torch.float32: 0, torch.bfloat16: 1}


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3


# (C, HW, x dtype, y dtype, gelu, dy dtype, dx dtype)
CASES = ((48, 112, torch.bfloat16, torch.bfloat16, 1, torch.bfloat16, torch.bfloat16), (96, 56, torch.bfloat16, torch.bfloat16, 0, torch.bfloat16, torch.bfloat16),
         (96, 56, torch.float32, torch.bfloat16, 0, torch.bfloat16, torch.float32), (192, 28, torch.float32, torch.bfloat16, 0, torch.bfloat16, torch.float32),
         (384, 14, torch.float32, torch.bfloat16, 0, torch.bfloat16, torch.float32), (768, 7, torch.float32, torch.bfloat16, 0, torch.bfloat16, torch.float32))
for C, HW, xdt, ydt, gelu, ddt, odt in CASES:
    M = B * HW * HW
    x = torch.randn(M, C, device="cuda").to(xdt)
    w = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    y = torch.empty(M, C, device="cuda", dtype=ydt)
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    t_f = timeit(lambda: lib.cnx_layernorm_fwd(x.data_ptr(), code[xdt], w.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), code[ydt], mean.data_ptr(), rstd.data_ptr(), M, C, gelu, S()))
    dy = torch.randn(M, C, device="cuda").to(ddt)
    dx = torch.empty(M, C, device="cuda", dtype=odt)
    dw = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device="cuda")
    t_b = timeit(lambda: lib.cnx_layernorm_bwd(dy.data_ptr(), code[ddt], x.data_ptr(), code[xdt], w.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx.data_ptr(), code[odt], dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, C, gelu, S()))
    t_n = timeit(lambda: lib.cnx_layernorm_bwd(dy.data_ptr(), code[ddt], x.data_ptr(), code[xdt], w.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx.data_ptr(), code[odt], None, None, None, M, C, gelu, S()))
    bf = M * C * (x.element_size() + y.element_size()); bb = M * C * (x.element_size() + dy.element_size() + dx.element_size())
    print(f"C={C:4d} HW={HW:3d} {str(xdt)[6:]:>8s}->{str(ydt)[6:]:<8s} fwd {t_f:7.1f} us ({bf / t_f / 1e6:5.2f} TB/s) | bwd dy {str(ddt)[6:]} dx {str(odt)[6:]} {t_b:7.1f} us ({bb / t_b / 1e6:5.2f} TB/s) | without sums {t_n:7.1f} us ({bb / t_n / 1e6:5.2f} TB/s)", flush=True)
