#!/usr/bin/env python3
"""One BASELINE configuration's AT step, timed (the `other_configs` leg of bench.py on its own, with a stack dump if it stalls).
Usage: python tools/cfg_step.py vit_b 224 256 2 [--graph 1] [--graph-train 1] [--steps 3] [--dump-after 120]"""
import argparse, faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("arch"); ap.add_argument("res", type=int); ap.add_argument("batch", type=int); ap.add_argument("n_iter", type=int)
ap.add_argument("--graph", type=int, default=1); ap.add_argument("--graph-train", type=int, default=1)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warm", type=int, default=4)
ap.add_argument("--dump-after", type=int, default=120)
a = ap.parse_args()
faulthandler.dump_traceback_later(a.dump_after, repeat=True, file=sys.stderr)
import torch
import revisiting_at_amd as R
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = R.get_new_model(a.arch, pretrained=False, not_original=True, img_size=a.res)
tr = R.ATTrainStep(model, a.arch, R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=a.n_iter, graph=a.graph), dev, lr=1e-3,
                   amp_dtype=torch.bfloat16, ema=True, graph_train=bool(a.graph and a.graph_train))
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(a.batch, 3, a.res, a.res, device=dev, generator=g)
y = torch.randint(0, 1000, (a.batch,), device=dev, generator=g)
for i in range(a.warm):
    t0 = time.perf_counter(); tr.step(x, y); torch.cuda.synchronize()
    print(f"warm {i}: {time.perf_counter() - t0:.2f} s", flush=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"{a.arch} res {a.res} batch {a.batch} n_iter {a.n_iter} graph {a.graph}/{a.graph_train} gemm {R.ops._GEMM_MODE}: "
      f"{a.batch / dt:.1f} img/s, {dt * 1e3:.2f} ms/step; attack graphs {R.graphed.STATS}; train graphs "
      f"{sum(v is not None for v in tr._tg.values())} ok / {sum(v is None for v in tr._tg.values())} failed", flush=True)
