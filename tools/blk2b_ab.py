#!/usr/bin/env python3
"""A/B in ONE process (cnx_runtime_switch 3): the Hpre backward of the fused LN+MLP block on one wavefront per row tile
(blk_mlp_bwd_kernel<C, ., ., ., true>) against the wavefront-pair kernel (blk2_bwd_kernel, round 6) at C = 384 / 256, in its three forms
(attack: du; training: da + dO rows + dHpre tiles; training with the LayerNorm backward in the epilogue).  Bit-equality of every
output, a spot check of du against autograd, then alternating timings.  usage: tools/blk2b_ab.py [C ...] [--batch B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import revisiting_at_amd as R
lib = R._lib.load()
dev = torch.device("cuda")
S = torch.cuda.current_stream().cuda_stream
SW = 3                                                      # CNX_SWITCH_BLK2_BWD_WIDTHS
args = [a for a in sys.argv[1:] if not a.startswith("--") and a.isdigit() and int(a) in (192, 256, 384)]
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 256


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]


for C in [int(v) for v in args] or [384, 256]:
    hw = {384: 14, 256: 28, 192: 28}[C]
    M = batch * hw * hw
    g = torch.Generator(device=dev).manual_seed(C)
    u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev, generator=g)
    w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
    lw, lb = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    b1, b2 = 0.1 * torch.randn(4 * C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    gm = 0.5 + 0.1 * torch.randn(C, device=dev, generator=g)
    gout = torch.randn(M, C, device=dev, generator=g)
    wf, wb = R.ops._pack_mlp(w1, w2), R.ops._pack_mlp_bwd(w1, w2)
    n_ws = lib.cnx_block_mlp_hpre_elems(M, C)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    out = torch.empty(M, C, device=dev)
    hp = torch.empty(n_ws, device=dev, dtype=torch.bfloat16)
    R._lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                            b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, hp.data_ptr(), M, C, S), "fwd_hpre")

    def forms(du, da, dos, dhp):
        return {
            "attack (du)": lambda: R._lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0,
                                                                                 gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), du.data_ptr(), M, C, S), "bwd_hpre"),
            "train (da, dO, dHpre)": lambda: R._lib.check(lib.cnx_block_mlp_bwd_train_hpre(gout.data_ptr(), 0, gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), da.data_ptr(),
                                                                                            dos.data_ptr(), dhp.data_ptr(), M, C, S), "bwd_train_hpre"),
            "train + LN (du, dO, dHpre)": lambda: R._lib.check(lib.cnx_block_mlp_bwd_train_hpre_ln(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                                                                   gout.data_ptr(), 0, gm.data_ptr(), wb.data_ptr(), hp.data_ptr(),
                                                                                                   du.data_ptr(), dos.data_ptr(), dhp.data_ptr(), M, C, S), "bwd_train_hpre_ln"),
        }
    ON = 7
    bufs = {w: dict(du=torch.zeros(M, C, device=dev, dtype=torch.bfloat16), da=torch.zeros(M, C, device=dev, dtype=torch.bfloat16),
                    dos=torch.zeros(M, C, device=dev, dtype=torch.bfloat16), dhp=torch.zeros(n_ws, device=dev, dtype=torch.bfloat16)) for w in (0, ON)}
    rows = torch.randint(0, M, (2048,), device=dev, generator=g)
    ur = u[rows].float().requires_grad_()
    a = F.layer_norm(ur, (C,), lw, lb, 1e-6)
    h = F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1)
    y = (h @ w2.to(torch.bfloat16).float().t() + b2) * gm
    (gu,) = torch.autograd.grad(y, ur, gout[rows])
    for name in ("attack (du)", "train (da, dO, dHpre)", "train + LN (du, dO, dHpre)"):
        outs = {}
        for w in (0, ON):
            lib.cnx_runtime_switch(SW, w)
            for t in bufs[w].values():
                t.zero_()
            forms(**bufs[w])[name]()
            torch.cuda.synchronize()
            outs[w] = {k: v.clone() for k, v in bufs[w].items()}
        same = {k: bool(torch.equal(outs[0][k], outs[ON][k])) for k in outs[0]}
        errs = {w: round(float((outs[w]["du"][rows].float() - gu).norm() / gu.norm()), 5) for w in (0, ON)} if "du" in name else {}
        ts = {0: [], ON: []}
        for rep in range(3):
            for w in (0, ON):
                lib.cnx_runtime_switch(SW, w)
                ts[w].append(round(timeit(forms(**bufs[w])[name]), 1))
        fl = 16.0 * M * C * C
        print(f"C={C} M={M} {name:28s} one wavefront / tile {ts[0]} us | pair {ts[ON]} us | ratio {sorted(ts[ON])[1] / sorted(ts[0])[1]:.3f} | "
              f"{fl / sorted(ts[ON])[1] / 1e6 / 2500:.3f} of MFMA peak (pair) | bit-equal {same} | du vs autograd {errs}", flush=True)
lib.cnx_runtime_switch(SW, 3)
