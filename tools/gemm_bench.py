#!/usr/bin/env python3
"""cnx_gemm_nt vs the library GEMM (+ the one-pass kernels it replaces) at the shapes of the models.  Usage: python tools/gemm_bench.py"""
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
    ts.sort(); return ts[len(ts) // 2], ts[0]


g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = [("cnx-T C=384 fc1", 50176, 1536, 384), ("cnx-T C=384 fc2", 50176, 384, 1536), ("cnx-T C=768 fc1", 12544, 3072, 768),
          ("cnx-T C=768 fc2", 12544, 768, 3072), ("downsample 96->192", 200704, 192, 384), ("downsample 192->384", 50176, 384, 768),
          ("downsample 384->768", 12544, 768, 1536), ("ViT-B qkv", 50432, 2304, 768), ("ViT-B proj", 50432, 768, 768),
          ("ViT-B fc1", 50432, 3072, 768), ("ViT-B fc2", 50432, 768, 3072), ("cnx-L C=1536 fc1 @10x10", 12800, 6144, 1536)]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    bb = b.bfloat16()
    fl = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.addmm(bb, a, w.t()))
    t_hip = timeit(lambda: O._gemm_nt(a, w, O.EPI_BIAS, bias=b))
    t_gelu = timeit(lambda: O._gemm_nt(a, w, O.EPI_BIAS_GELU, bias=b))
    t_libg = timeit(lambda: O._gelu_bf16(torch.addmm(bb, a, w.t())))
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d} | bias: hip {t_hip[0]:7.1f} us ({fl / t_hip[0] / 1e6:6.0f} TF) lib {t_lib[0]:7.1f} us ({fl / t_lib[0] / 1e6:6.0f} TF)"
          f" | bias+GELU: hip {t_gelu[0]:7.1f} us  lib+gelu {t_libg[0]:7.1f} us", flush=True)
