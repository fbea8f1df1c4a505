#!/usr/bin/env python3
"""Host-side (Python) profile of the AT step: where the ~50 ms of enqueue time per step go.  Usage: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R

dev = torch.device("cuda")
torch.manual_seed(0)
model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True)
tr = R.ATTrainStep(model, "convnext_tiny", R.AdvConfig(attack="apgd", n_iter=2), dev, lr=1e-3, gemm_table=True)
x = torch.rand(256, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (256,), device=dev)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, total {1e3 * (t2 - t0) / 5:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(45)
