#!/usr/bin/env python3
"""A/B in ONE process (cnx_runtime_switch 0): the fused LN+MLP forward at C = 192 on one wavefront per tile (two workgroups per CU)
against the wavefront-pair kernel (blk2_fwd_kernel<192>, one workgroup per CU), plain / Hpre / training forms; bit-equality of every output, then
alternating timings (median of 20 per leg, 3 legs each).  usage: tools/w8_ab.py [C ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
dev = torch.device("cuda")
S = torch.cuda.current_stream().cuda_stream
SW = 0                                                      # CNX_SWITCH_BLK2_WIDTHS


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]


for C in [int(v) for v in sys.argv[1:]] or [192]:
    hw = {192: 28, 128: 56, 96: 56, 256: 28}[C]
    M = 256 * hw * hw
    g = torch.Generator(device=dev).manual_seed(C)
    u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev, generator=g)
    w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
    lw, lb = 1 + 0.1 * torch.randn(C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    b1, b2 = 0.1 * torch.randn(4 * C, device=dev, generator=g), 0.1 * torch.randn(C, device=dev, generator=g)
    gm = 0.5 + 0.1 * torch.randn(C, device=dev, generator=g)
    wf = R.ops._pack_mlp(w1, w2)
    n_ws = lib.cnx_block_mlp_hpre_elems(M, C)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)

    def forms(out, hp, hact, arows, y2):
        return {
            "fwd": lambda: R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                                              b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, None, M, C, S), "fwd"),
            "fwd_hpre": lambda: R._lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(),
                                                                        wf.data_ptr(), b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(),
                                                                        0, hp.data_ptr(), M, C, S), "fwd_hpre"),
            "fwd_train": lambda: R._lib.check(lib.cnx_block_mlp_fwd_train(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(),
                                                                          wf.data_ptr(), b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(),
                                                                          0, y2.data_ptr(), hp.data_ptr(), hact.data_ptr(), arows.data_ptr(), M, C, S), "fwd_train"),
        }
    bufs = {}
    for w in (3, 7):
        bufs[w] = dict(out=torch.zeros(M, C, device=dev), hp=torch.zeros(n_ws, device=dev, dtype=torch.bfloat16),
                       hact=torch.zeros(n_ws, device=dev, dtype=torch.bfloat16), arows=torch.zeros(M, C, device=dev, dtype=torch.bfloat16),
                       y2=torch.zeros(M, C, device=dev, dtype=torch.bfloat16))
    for name in ("fwd", "fwd_hpre", "fwd_train"):
        outs = {}
        for w in (3, 7):
            lib.cnx_runtime_switch(SW, w)
            b = bufs[w]
            for t in b.values():
                t.zero_()
            forms(**b)[name]()
            torch.cuda.synchronize()
            outs[w] = {k: v.clone() for k, v in b.items()}
        same = {k: bool(torch.equal(outs[3][k], outs[7][k])) for k in outs[3]}
        ts = {3: [], 7: []}
        for rep in range(3):
            for w in (3, 7):
                lib.cnx_runtime_switch(SW, w)
                ts[w].append(round(timeit(forms(**bufs[w])[name]), 1))
        fl = 16.0 * M * C * C
        print(f"C={C} M={M} {name:10s} one wavefront / tile {ts[3]} us | pair {ts[7]} us | ratio {sorted(ts[7])[1] / sorted(ts[3])[1]:.3f} | "
              f"{fl / sorted(ts[7])[1] / 1e6 / 2500:.3f} of MFMA peak (pair) | bit-equal {same}", flush=True)
lib.cnx_runtime_switch(SW, 3)
