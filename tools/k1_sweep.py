#!/usr/bin/env python3
"""Launch-shape sweep of the APGD kernels at BASELINE config #2 size (B=256, 3x224x224).
Prints one line per variant: achieved algorithmic GB/s (20 B/elem for the Linf step)."""
import itertools
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R

lib = R._lib.load()
B, E = 256, 3 * 224 * 224
eps = 4 / 255
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(B, E, device="cuda", generator=g)
xa = (x + (torch.rand(B, E, device="cuda", generator=g) * 2 - 1) * eps).clamp(0, 1)
xo = (x + (torch.rand(B, E, device="cuda", generator=g) * 2 - 1) * eps).clamp(0, 1)
gr = torch.randn(B, E, device="cuda", generator=g) * 1e-3
grb = gr.to(torch.bfloat16)
step = torch.full((B,), 2 * eps, device="cuda")
out = torch.empty_like(x)
outb = torch.empty(B, E, device="cuda", dtype=torch.bfloat16)
S = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


N = B * E
print("# copy baseline (torch clone: 8 B/elem)")
med, mn = timeit(lambda: out.copy_(x))
print(f"copy                           med {med*1e3:8.1f} us  {8*N/med/1e6:8.1f} GB/s   min {8*N/mn/1e6:8.1f}")
for bps, un, nt in itertools.product([37, 74, 147], [1, 2, 4], [0, 1]):
    fn = lambda: lib.apgd_linf_step_f32_ex(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), gr.data_ptr(), 0, step.data_ptr(),
                                            out.data_ptr(), None, B, E, eps, 0.75, bps, un, nt, S)
    med, mn = timeit(fn)
    print(f"linf f32  bps={bps:4d} U={un} nt={nt}  med {med*1e3:8.1f} us  {20*N/med/1e6:8.1f} GB/s   min-time {20*N/mn/1e6:8.1f} GB/s")
for bps, un, nt in itertools.product([8, 16, 37], [2, 4], [0, 1]):
    fn = lambda: lib.apgd_linf_step_f32_ex(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), grb.data_ptr(), 1, step.data_ptr(),
                                            out.data_ptr(), outb.data_ptr(), B, E, eps, 0.75, bps, un, nt, S)
    med, mn = timeit(fn)
    print(f"linf bf16g+bf16out bps={bps:4d} U={un} nt={nt}  med {med*1e3:8.1f} us  {20*N/med/1e6:8.1f} GB/s (20 B/elem: 12 rd + 2 rd + 4 wr + 2 wr)")
sg = torch.sign(gr).to(torch.int8)
for bps, un, nt in itertools.product([37, 74, 147], [1, 2, 4], [0, 1]):
    fn = lambda: lib.apgd_linf_step_f32_ex(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), sg.data_ptr(), 3, step.data_ptr(),
                                            out.data_ptr(), None, B, E, eps, 0.75, bps, un, nt, S)
    med, mn = timeit(fn)
    print(f"linf int8-sign bps={bps:4d} U={un} nt={nt}  med {med*1e3:8.1f} us  algorithmic(20B) {20*N/med/1e6:8.1f} GB/s   moved(17B) {17*N/med/1e6:8.1f} GB/s")
for code, gt, nm, mv in ((0, gr, "f32", 16), (3, sg, "int8", 13)):
    fn = lambda: lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xa.data_ptr(), gt.data_ptr(), code, step.data_ptr(),
                                         out.data_ptr(), None, B, E, eps, 1.0, S)
    med, mn = timeit(fn)
    print(f"linf first-iter form, {nm} grad  med {med*1e3:8.1f} us  algorithmic(16B) {16*N/med/1e6:8.1f} GB/s   moved({mv}B) {mv*N/med/1e6:8.1f} GB/s")
xb, xba = torch.empty_like(x), torch.empty_like(x)
med, mn = timeit(lambda: lib.apgd_init_f32(x.data_ptr(), out.data_ptr(), xb.data_ptr(), xba.data_ptr(), N, S))
print(f"init (4 rd + 12 wr)            med {med*1e3:8.1f} us  {16*N/med/1e6:8.1f} GB/s")
flags = torch.full((B,), 3, device="cuda", dtype=torch.uint8)
gb = torch.empty_like(gr)
med, mn = timeit(lambda: lib.apgd_track_rows(flags.data_ptr(), out.data_ptr(), gr.data_ptr(), xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), 4, B, E, 0, S))
print(f"track all rows flags=3 (8 rd + 12 wr) med {med*1e3:8.1f} us  {20*N/med/1e6:8.1f} GB/s")
