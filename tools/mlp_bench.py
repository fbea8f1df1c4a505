#!/usr/bin/env python3
"""Micro-benchmark + spot check of the fused LN+MLP block kernels through the C ABI (cnx_block_mlp_fwd / _bwd_input / _bwd).
Kernel variants are selected by environment variables read once per process (APGD_BLK_FWD_IMPL, APGD_MLP2_RG, ...), so run
it once per variant.  Usage: python tools/mlp_bench.py [--C 96] [--hw 56] [--batch 256] [--what fwd,bwd_in,bwd] [--tag x]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import revisiting_at_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=96)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--what", default="fwd")
ap.add_argument("--tag", default="")
ap.add_argument("--resid", default="f32")
ap.add_argument("--check-rows", type=int, default=4096)
args = ap.parse_args()
lib = R._lib.load()
dev = torch.device("cuda")
C, M = args.C, args.batch * args.hw * args.hw
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
x = torch.randn(M, C, device=dev, generator=g)
if args.resid == "bf16":
    x = x.to(torch.bfloat16)
w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
lw = 1 + 0.1 * torch.randn(C, device=dev, generator=g)
lb = 0.1 * torch.randn(C, device=dev, generator=g)
b1 = 0.1 * torch.randn(4 * C, device=dev, generator=g)
b2 = 0.1 * torch.randn(C, device=dev, generator=g)
gm = 0.5 + 0.1 * torch.randn(C, device=dev, generator=g)
code = R._lib.dtype_code


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def ref_fwd(rows):
    a = F.layer_norm(u[rows].float(), (C,), lw, lb, 1e-6).to(torch.bfloat16).float()
    h = F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1).to(torch.bfloat16).float()
    y = h @ w2.to(torch.bfloat16).float().t() + b2
    return x[rows].float() + gm * y


res = {"tag": args.tag, "C": C, "M": M, "env": {k: v for k, v in os.environ.items() if k.startswith("APGD_")}}
flops = 16.0 * M * C * C
if "fwd" in args.what.split(","):
    wf = R.ops._pack_mlp(w1, w2)
    out = torch.empty(M, C, device=dev)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)

    def run():
        R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                           b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), code(x.dtype), out.data_ptr(), 0, None, M, C,
                                           S), "fwd")
    run()
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, args.check_rows, device=dev), torch.arange(M - args.check_rows, M, device=dev),
                      torch.randint(0, M, (args.check_rows,), device=dev, generator=g)])
    want = ref_fwd(rows)
    err = float((out[rows] - want).abs().max() / want.abs().max())
    mu = u[rows].float().mean(1)
    merr = float((mean[rows] - mu).abs().max())
    med, mn = timeit(run)
    nbytes = M * C * (2 + x.element_size() + 4)
    import hashlib
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16] + "/" + hashlib.sha256(mean.cpu().numpy().tobytes()).hexdigest()[:8]
    res["fwd_out_sha"] = sha
    res["fwd"] = {"med_us": round(med, 1), "min_us": round(mn, 1), "TFLOPs": round(flops / med / 1e6, 1), "mfma_frac": round(flops / med / 1e6 / 2500, 4),
                  "GBs": round(nbytes / med / 1e3, 1), "max_rel_err": err, "mean_err": merr}
if "bwd_in" in args.what.split(","):
    wf = R.ops._pack_mlp(w1, w2)
    wb = R.ops._pack_mlp_bwd(w1, w2)
    out = torch.empty(M, C, device=dev)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                       b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), code(x.dtype), out.data_ptr(), 0, None, M, C, S),
                 "fwd")
    gout = torch.randn(M, C, device=dev, generator=g)
    du = torch.empty(M, C, device=dev, dtype=torch.bfloat16)

    def runb():
        R._lib.check(lib.cnx_block_mlp_bwd_input(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0,
                                                 gm.data_ptr(), wb.data_ptr(), b1.data_ptr(), du.data_ptr(), M, C, S), "bwd_input")
    runb()
    torch.cuda.synchronize()
    # spot check against autograd of the bf16-rounded reference
    rows = torch.randint(0, M, (min(args.check_rows, 2048),), device=dev, generator=g)
    ur = u[rows].float().requires_grad_()
    a = F.layer_norm(ur, (C,), lw, lb, 1e-6)
    h = F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1)
    y = (h @ w2.to(torch.bfloat16).float().t() + b2) * gm
    (gu,) = torch.autograd.grad(y, ur, gout[rows])
    err = float((du[rows].float() - gu).norm() / gu.norm())
    med, mn = timeit(runb)
    res["bwd_in"] = {"med_us": round(med, 1), "min_us": round(mn, 1), "TFLOPs": round(1.5 * flops / med / 1e6, 1),
                     "mfma_frac": round(1.5 * flops / med / 1e6 / 2500, 4), "rel_err": err}
if "hpre" in args.what.split(","):
    wf = R.ops._pack_mlp(w1, w2)
    wb = R.ops._pack_mlp_bwd(w1, w2)
    out = torch.empty(M, C, device=dev)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    hp = torch.empty(lib.cnx_block_mlp_hpre_elems(M, C), device=dev, dtype=torch.bfloat16)
    gout = torch.randn(M, C, device=dev, generator=g)
    du = torch.empty(M, C, device=dev, dtype=torch.bfloat16)

    def runf():
        R._lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                                b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), code(x.dtype), out.data_ptr(), 0,
                                                hp.data_ptr(), M, C, S), "fwd_hpre")

    def runb():
        R._lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0,
                                                      gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), du.data_ptr(), M, C, S), "bwd_hpre")
    runf()
    runb()
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (min(args.check_rows, 2048),), device=dev, generator=g)
    ur = u[rows].float().requires_grad_()
    a = F.layer_norm(ur, (C,), lw, lb, 1e-6)
    h = F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1)
    y = (h @ w2.to(torch.bfloat16).float().t() + b2) * gm
    (gu,) = torch.autograd.grad(y, ur, gout[rows])
    err = float((du[rows].float() - gu).norm() / gu.norm())
    want = ref_fwd(rows)
    ferr = float((out[rows] - want).abs().max() / want.abs().max())
    medf, mnf = timeit(runf)
    medb, mnb = timeit(runb)
    res["hpre"] = {"fwd_med_us": round(medf, 1), "fwd_min_us": round(mnf, 1), "bwd_med_us": round(medb, 1), "bwd_min_us": round(mnb, 1),
                   "fwd_max_rel_err": ferr, "bwd_rel_err": err}
print(json.dumps(res))
