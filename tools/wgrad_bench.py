#!/usr/bin/env python3
"""Split-K factor sweep for the weight-gradient GEMMs of the fused stages (ops._wgrad_t): time of bmm + partial sum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
from revisiting_at_amd import ops

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for N1, M, N2 in ((384, 802816, 96), (768, 200704, 192)):
    xt = torch.randn(N1, M, device="cuda").to(torch.bfloat16)
    y = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    row = [f"N1={N1} M={M} N2={N2}:"]
    for S in (32, 64, 128, 256, 512, 1024):
        if M % S: continue
        def f():
            part = torch.bmm(xt.view(N1, S, M // S).transpose(0, 1), y.view(S, M // S, N2))
            return ops._sum_parts(part)
        row.append(f"S={S}: {t(f):7.1f} us")
    print("  ".join(row))

for M, N1, N2 in ((50176, 384, 1536), (50176, 1536, 384), (12544, 768, 3072), (12544, 3072, 768), (200704, 192, 384), (50176, 384, 768), (12544, 768, 1536)):
    x = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    y = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    row = [f"M={M} N1={N1} N2={N2}:"]
    row.append(f"S=1: {t(lambda: (x.t() @ y).float()):7.1f} us")
    for S in (2, 4, 8, 16, 32, 64, 128):
        if M % S: continue
        def f():
            part = torch.bmm(x.view(S, M // S, N1).transpose(1, 2), y.view(S, M // S, N2))
            return ops._sum_parts(part)
        row.append(f"S={S}: {t(f):7.1f} us")
    print("  ".join(row))
