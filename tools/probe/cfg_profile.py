#!/usr/bin/env python3
"""One of the other BASELINE configurations in isolation for a kernel profile.
Usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/probe/cfg_profile.py vit_b 224 256 2 [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R
arch, res, batch, n_iter = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device("cuda")
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True                       # as bench.py (main.py:25)
R.ops.load_gemm_table()                                     # as bench.py's trainer (APGD_GEMM_TABLE=0 disables)
model = R.get_new_model(arch, pretrained=False, not_original=True, img_size=res)
tr = R.ATTrainStep(model, arch, R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=n_iter), dev, lr=1e-3, amp_dtype=torch.bfloat16, ema=True)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(batch, 3, res, res, device=dev, generator=g)
y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(x, y)
te = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("ms per step", dt / steps * 1e3, "host enqueue ms", te / steps * 1e3)
