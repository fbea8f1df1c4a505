#!/usr/bin/env python3
"""The pieces of a ConvNeXt block WITHOUT a fused LN+MLP kernel (C = 768: stage 3 of ConvNeXt-T; C >= 512 of -B / -L) as the attack and the
training pass run them: LayerNorm, cnx_gemm_nt with each epilogue, against the plain library GEMM of the same shape.  us per launch
(median of 20), isolated.  usage: tools/c768_bench.py [C] [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
O = R.ops
lib = R._lib.load()


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]


C = int(sys.argv[1]) if len(sys.argv) > 1 else 768
Ms = [int(v) for v in sys.argv[2:]] or [12544, 6272]
g = torch.Generator(device="cuda").manual_seed(0)
S = lambda: torch.cuda.current_stream().cuda_stream
for M in Ms:
    u = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    x = torch.randn(M, C, device="cuda", generator=g)
    w1 = (torch.randn(4 * C, C, device="cuda", generator=g) * C ** -0.5).bfloat16()
    w2 = (torch.randn(C, 4 * C, device="cuda", generator=g) * (4 * C) ** -0.5).bfloat16()
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1, b2, gm = torch.randn(4 * C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g)
    lw, lb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    a = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    hpre = torch.empty(M, 4 * C, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * 4 * C * C
    t = {}
    t["LayerNorm fwd (bf16 -> bf16)"] = timeit(lambda: lib.cnx_layernorm_fwd(u.data_ptr(), 1, lw.data_ptr(), lb.data_ptr(), 1e-6, a.data_ptr(), 1,
                                                                              mean.data_ptr(), rstd.data_ptr(), M, C, 0, S()))
    h = O._gemm_nt(a, w1, O.EPI_BIAS_GELU, bias=b1, z_out=hpre)
    t["fc1 plain bias (no GELU)"] = timeit(lambda: O._gemm_nt(a, w1, O.EPI_BIAS, bias=b1))
    t["fc1 + bias + GELU"] = timeit(lambda: O._gemm_nt(a, w1, O.EPI_BIAS_GELU, bias=b1))
    t["fc1 + bias + GELU + Hpre out"] = timeit(lambda: O._gemm_nt(a, w1, O.EPI_BIAS_GELU, bias=b1, z_out=hpre))
    t["fc1 library addmm"] = timeit(lambda: torch.addmm(b1.bfloat16(), a, w1.t()))
    t["fc2 plain bias"] = timeit(lambda: O._gemm_nt(h, w2, O.EPI_BIAS, bias=b2))
    t["fc2 + bias + gamma + fp32 residual -> fp32"] = timeit(lambda: O._gemm_nt(h, w2, O.EPI_SCALE_RES, bias=b2, gamma=gm, resid=x, out_dtype=torch.float32))
    xb = x.bfloat16()
    t["fc2 + bias + gamma + bf16 residual -> fp32"] = timeit(lambda: O._gemm_nt(h, w2, O.EPI_SCALE_RES, bias=b2, gamma=gm, resid=xb, out_dtype=torch.float32))
    t["fc2 library addmm"] = timeit(lambda: torch.addmm(b2.bfloat16(), h, w2.t()))
    dos = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    t["dH = dO W2 with GELU' (epi 3)"] = timeit(lambda: O._gemm_nt(dos, w2t, O.EPI_GELU_GRAD, z_in=hpre))
    t["dH = dO W2 plain"] = timeit(lambda: O._gemm_nt(dos, w2t, O.EPI_BIAS))
    dhp = O._gemm_nt(dos, w2t, O.EPI_GELU_GRAD, z_in=hpre)
    t["da = dHpre W1 plain"] = timeit(lambda: O._gemm_nt(dhp, w1t, O.EPI_BIAS))
    print(f"--- C = {C}, M = {M}: one GEMM = {fl / 1e9:.1f} GFLOP")
    for k, v in t.items():
        extra = f"  {fl / v / 1e6:6.0f} TF/s" if ("fc" in k or "dH" in k or "da" in k) else ""
        print(f"  {k:46s} {v:8.1f} us{extra}", flush=True)
