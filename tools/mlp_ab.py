#!/usr/bin/env python3
"""Interleaved A/B timing of two builds of the library on the fused LN+MLP kernels (same process, same tensors, alternating
launches: box-to-box and minute-to-minute drift cancels).  Usage: python tools/mlp_ab.py libA.so libB.so [--C 192,384] [--rounds 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import revisiting_at_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--C", default="96,192,384")
ap.add_argument("--rounds", type=int, default=30)
args = ap.parse_args()
import ctypes


def load_partial(path):
    """Only the entry points used here (an older build may lack newer symbols)."""
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("cnx_mlp_packed_elems", "cnx_mlp_packed_bwd_elems", "cnx_mlp_pack_weights", "cnx_mlp_pack_weights_bwd",
                 "cnx_block_mlp_hpre_elems", "cnx_block_mlp_fwd", "cnx_block_mlp_fwd_hpre", "cnx_block_mlp_bwd_input_hpre",
                 "cnx_block_mlp_bwd_input", "cnx_block_mlp_hpre_supported", "cnx_block_mlp_bwd_supported"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = R._lib.PROTOTYPES[name]
    return lib


libs = [load_partial(p) for p in args.libs]
dev = torch.device("cuda")
S = torch.cuda.current_stream().cuda_stream
code = R._lib.dtype_code


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


for C in [int(c) for c in args.C.split(",")]:
    hw = 5376 // C
    M = 256 * hw * hw
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev, generator=g)
    w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
    lw = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
    lb = 0.1 * torch.randn(C, device=dev, generator=g)
    b1 = 0.1 * torch.randn(4 * C, device=dev, generator=g)
    b2 = 0.1 * torch.randn(C, device=dev, generator=g)
    gm = 0.5 + 0.1 * torch.randn(C, device=dev, generator=g)
    gout = torch.randn(M, C, device=dev, generator=g)
    out = torch.empty(M, C, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    du = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    fns = {}
    for li, lib in enumerate(libs):
        wf = torch.empty(lib.cnx_mlp_packed_elems(C), device=dev, dtype=torch.bfloat16)
        wb = torch.empty(lib.cnx_mlp_packed_bwd_elems(C), device=dev, dtype=torch.bfloat16)
        R._lib.check(lib.cnx_mlp_pack_weights(w1.data_ptr(), w2.data_ptr(), 0, wf.data_ptr(), C, S), "pack")
        R._lib.check(lib.cnx_mlp_pack_weights_bwd(w1.data_ptr(), w2.data_ptr(), 0, wb.data_ptr(), C, S), "packb")
        hp = torch.empty(max(1, lib.cnx_block_mlp_hpre_elems(M, C)), device=dev, dtype=torch.bfloat16)

        def fwd(lib=lib, wf=wf):
            R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                               b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, None, M, C, S), "fwd")

        def hfwd(lib=lib, wf=wf, hp=hp):
            R._lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(),
                                                    wf.data_ptr(), b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0,
                                                    hp.data_ptr(), M, C, S), "hfwd")

        def hbwd(lib=lib, wb=wb, hp=hp):
            R._lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0,
                                                          gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), du.data_ptr(), M, C, S), "hbwd")

        def bwd_in(lib=lib, wb=wb):
            R._lib.check(lib.cnx_block_mlp_bwd_input(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0,
                                                     gm.data_ptr(), wb.data_ptr(), b1.data_ptr(), du.data_ptr(), M, C, S), "bwd_in")
        fns[li] = {"fwd": fwd}
        if lib.cnx_block_mlp_hpre_supported(C):
            fns[li].update(hfwd=hfwd, hbwd=hbwd)
        if lib.cnx_block_mlp_bwd_supported(C):
            fns[li]["bwd_in"] = bwd_in
    for name in fns[0]:
        if name not in fns[1]:
            continue
        for f in (fns[0][name], fns[1][name]):
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        ts = ([], [])
        for r in range(args.rounds):
            for li in ((0, 1) if r % 2 == 0 else (1, 0)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fns[li][name](); e1.record(); e1.synchronize()
                ts[li].append(e0.elapsed_time(e1) * 1e3)
        a, b = median(ts[0]), median(ts[1])
        print(f"C={C:4d} {name:7s} A {a:7.1f} us (min {min(ts[0]):7.1f})  B {b:7.1f} us (min {min(ts[1]):7.1f})  B/A {b / a:.3f}", flush=True)
