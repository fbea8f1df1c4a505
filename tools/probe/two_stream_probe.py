#!/usr/bin/env python3
"""Does running the C = 384 fused forward as two half-batch chains on two streams fill the idle CUs of its second round
(392 workgroups of 128 rows on 256 CUs = 1.53 rounds)?  Chains of `depth` dependent launches (out -> next residual)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import revisiting_at_amd as R

lib = R._lib.load()
dev = torch.device("cuda")
code = R._lib.dtype_code


def setup(C, M):
    g = torch.Generator(device=dev).manual_seed(0)
    d = dict(u=torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16), x=torch.randn(M, C, device=dev, generator=g),
             out=torch.empty(M, C, device=dev), mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev))
    return d


def params(C):
    g = torch.Generator(device=dev).manual_seed(1)
    w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
    return dict(wf=R.ops._pack_mlp(w1, w2), lw=torch.ones(C, device=dev), lb=torch.zeros(C, device=dev),
                b1=torch.zeros(4 * C, device=dev), b2=torch.zeros(C, device=dev), gm=torch.ones(C, device=dev))


def launch(p, d, M, C, stream, lo=0, n=None):
    n = M if n is None else n
    sl = lambda t: t[lo:lo + n]
    R._lib.check(lib.cnx_block_mlp_fwd(sl(d["u"]).data_ptr(), p["lw"].data_ptr(), p["lb"].data_ptr(), 1e-6, sl(d["mean"]).data_ptr(),
                                       sl(d["rstd"]).data_ptr(), p["wf"].data_ptr(), p["b1"].data_ptr(), p["b2"].data_ptr(),
                                       p["gm"].data_ptr(), sl(d["x"]).data_ptr(), code(d["x"].dtype), sl(d["out"]).data_ptr(), 0, None, n, C,
                                       stream.cuda_stream), "fwd")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for C, M in ((384, 50176), (192, 200704), (96, 802816), (384, 25088 * 3)):
    p, d = params(C), setup(C, M)
    main = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    depth = 9

    def one():
        for _ in range(depth):
            launch(p, d, M, C, main)

    def two():
        ev = torch.cuda.Event()
        ev.record(main)
        s1.wait_event(ev)
        s2.wait_event(ev)
        h = (M // 2 + 127) // 128 * 128
        for _ in range(depth):
            launch(p, d, M, C, s1, 0, h)
            launch(p, d, M, C, s2, h, M - h)
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1)
        e2.record(s2)
        main.wait_event(e1)
        main.wait_event(e2)

    def three():                                  # main stream + one side stream: 2/3 - 1/3?  no: thirds on three streams
        ev = torch.cuda.Event()
        ev.record(main)
        s1.wait_event(ev)
        s2.wait_event(ev)
        t = (M // 3 + 127) // 128 * 128
        for _ in range(depth):
            launch(p, d, M, C, main, 0, t)
            launch(p, d, M, C, s1, t, t)
            launch(p, d, M, C, s2, 2 * t, M - 2 * t)
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1)
        e2.record(s2)
        main.wait_event(e1)
        main.wait_event(e2)

    a, b, c = timed(one), timed(two), timed(three)
    print(f"C={C} M={M}: chain of {depth}: one stream {a / depth:.1f} us/launch-equivalent, two half-batch streams {b / depth:.1f}, "
          f"three third-batch streams {c / depth:.1f}", flush=True)
