#!/usr/bin/env python3
"""Fused MLP forward (cnx_mlp_fwd) vs the eager hipBLASLt composition at ConvNeXt-T stage shapes, B=256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import revisiting_at_amd as R
lib = R._lib.load()
S = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]


for C, HW in ((96, 56), (192, 28), (384, 14)):
    M = 256 * HW * HW
    a = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, C, device="cuda")
    w1 = (torch.randn(4 * C, C, device="cuda") * C ** -0.5).to(torch.bfloat16)
    w2 = (torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5).to(torch.bfloat16)
    b1, b2, gm = torch.randn(4 * C, device="cuda"), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    w2p = w2[:, R.ops._w2_perm(4 * C, "cuda")].contiguous()
    out = torch.empty(M, C, device="cuda")
    b1b, b2b = b1.to(torch.bfloat16), b2.to(torch.bfloat16)
    fused = lambda: lib.cnx_mlp_fwd(a.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gm.data_ptr(),
                                    x.data_ptr(), 0, out.data_ptr(), 0, None, M, C, S)
    eager = lambda: x + F.linear(F.gelu(F.linear(a, w1, b1b)), w2, b2b) * gm
    tf, te = timeit(fused), timeit(eager)
    flops = 2 * 2 * M * C * 4 * C
    print(f"C={C:4d} M={M:7d}: fused {tf*1e3:8.1f} us ({flops/tf/1e9:7.1f} TFLOP/s, {(M*C*(2+4+4))/tf/1e6:7.1f} GB/s alg)   "
          f"eager {te*1e3:8.1f} us   speedup {te/tf:.2f}x")
