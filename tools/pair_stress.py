#!/usr/bin/env python3
"""Race hunt for the wavefront-pair kernels (LDS hand-overs, barriers, hand-counted DMA waits): every form of the forward and of the Hpre backward
N times on the same inputs, each output compared bit for bit with the first run's (and with the single-wavefront kernel's).
usage: tools/pair_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
dev = torch.device("cuda")
S = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for C, M in ((384, 50176), (384, 25088 + 96), (256, 200704), (256, 6272 + 32)):
    g = torch.Generator(device=dev).manual_seed(C + M)
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
    mk = lambda *sh, dt=torch.bfloat16: torch.zeros(sh, device=dev, dtype=dt)
    out, hp, hact, arows, y2 = mk(M, C, dt=torch.float32), mk(n_ws), mk(n_ws), mk(M, C), mk(M, C)
    du, da, dos, dhp = mk(M, C), mk(M, C), mk(M, C), mk(n_ws)

    def fwd_train():
        R._lib.check(lib.cnx_block_mlp_fwd_train(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(),
                                                 b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, y2.data_ptr(), hp.data_ptr(), hact.data_ptr(),
                                                 arows.data_ptr(), M, C, S), "fwd_train")
        return (out, hp, hact, arows, y2)

    def fwd_plain():
        R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(),
                                           b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, None, M, C, S), "fwd")
        return (out,)

    def bwd_attack():
        R._lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0, gm.data_ptr(),
                                                      wb.data_ptr(), hp.data_ptr(), du.data_ptr(), M, C, S), "bwd")
        return (du,)

    def bwd_train_ln():
        R._lib.check(lib.cnx_block_mlp_bwd_train_hpre_ln(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gout.data_ptr(), 0, gm.data_ptr(),
                                                         wb.data_ptr(), hp.data_ptr(), du.data_ptr(), dos.data_ptr(), dhp.data_ptr(), M, C, S), "bwd_ln")
        return (du, dos, dhp)

    def bwd_train():
        R._lib.check(lib.cnx_block_mlp_bwd_train_hpre(gout.data_ptr(), 0, gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), da.data_ptr(), dos.data_ptr(),
                                                      dhp.data_ptr(), M, C, S), "bwd_train")
        return (da, dos, dhp)

    for name, fn in (("fwd_train", fwd_train), ("fwd", fwd_plain), ("bwd_attack", bwd_attack), ("bwd_train_ln", bwd_train_ln), ("bwd_train", bwd_train)):
        lib.cnx_runtime_switch(0, 0); lib.cnx_runtime_switch(3, 0)
        ref1 = [t.clone() for t in fn()]                     # single-wavefront kernels
        lib.cnx_runtime_switch(0, 3); lib.cnx_runtime_switch(3, 3)
        n_bad = 0
        for r in range(reps):
            for t in fn():
                t.add_(1)                                    # make sure the next launch really rewrites everything
            outs = fn()
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(outs, ref1)):
                n_bad += 1
        bad += n_bad
        print(f"C={C} M={M} {name:13s}: {reps} launches, {n_bad} differ from the single-wavefront kernel's result", flush=True)
    fwd_train()                                              # leave a valid workspace
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
