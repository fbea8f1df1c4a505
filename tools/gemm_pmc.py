#!/usr/bin/env python3
"""A few launches of cnx_gemm_nt at one shape (for rocprofv3 --pmc).  Usage: python tools/gemm_pmc.py M N K [epi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
O = R.ops
M, N, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda", generator=g)
for _ in range(6):
    O._gemm_nt(a, w, epi, bias=b)
    torch.addmm(b.bfloat16(), a, w.t())
torch.cuda.synchronize()
