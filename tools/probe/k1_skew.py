#!/usr/bin/env python3
"""Does skewing the base addresses of K1's five streams (x, x_adv, x_adv_old, int8 signs, out) against each other change its HBM rate?
(Same relative offset in every stream = same HBM channel at the same time.)  Usage: python tools/probe/k1_skew.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
B, E = 256, 3 * 224 * 224
n = B * E
S = torch.cuda.current_stream().cuda_stream
dev = "cuda"
pool = torch.empty(6 * n * 4 + (1 << 24), device=dev, dtype=torch.uint8)
step = torch.full((B,), 0.03, device=dev)
flush = torch.empty(1 << 28, device=dev)          # 1 GiB: evicts the 256 MB MALL between launches


def carve(offsets):
    ts = []
    base = 0
    for i, sz in enumerate((4, 4, 4, 1, 4)):
        o = base + offsets[i]
        o = (o + 255) // 256 * 256
        t = pool[o:o + n * sz]
        ts.append(t.view(torch.float32 if sz == 4 else torch.int8))
        base = o + n * sz
    return ts


def run(offsets, reps=20, cold=True):
    x, xa, xo, sg, out = carve(offsets)
    x.uniform_(0, 1); xa.copy_(x); xo.copy_(x); sg.random_(-1, 2)
    ts = []
    for _ in range(reps):
        if cold:
            flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), sg.data_ptr(), 3, step.data_ptr(), out.data_ptr(), None, B, E,
                                    4 / 255, 0.75, S)
        e1.record(); e1.synchronize()
        assert rc == 0
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for name, offs in (("no skew", (0, 0, 0, 0, 0)), ("256 B steps", (0, 256, 512, 768, 1024)), ("4 KiB + 256 B steps", (0, 4352, 8704, 13056, 17408)),
                   ("1 MiB steps", (0, 1 << 20, 2 << 20, 3 << 20, 4 << 20)), ("prime-ish", (0, 256 * 37, 256 * 71, 256 * 113, 256 * 151))):
    for cold in (True, False):
        med, mn = run(offs, cold=cold)
        print(f"{name:22s} {'cold MALL' if cold else 'back to back'}: median {med:6.1f} us  fastest {mn:6.1f} us   ({770.7e6 / med / 1e6:.2f} TB/s algorithmic)")
