#!/usr/bin/env python3
"""Launch-shape sweep of the APGD Linf update (K1) INSIDE the adversarial-training step: the kernel's operands were last touched a
whole model pass ago, so they come from HBM - isolated loops (tools/k1_sweep.py) re-read them out of the Infinity Cache and rank the
shapes differently.  Prints, per shape (blocks per sample, unroll, non-temporal), the HIP-event time of the general-form launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
from revisiting_at_amd import apgd as A

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True)
tr = R.ATTrainStep(model, "convnext_tiny", R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2, graph=1), dev, lr=1e-3,
                   amp_dtype=torch.bfloat16, ema=True, gemm_table=True)
g = torch.Generator(device=dev).manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand(B, 3, 224, 224, device=dev, generator=g)
y = torch.randint(0, 1000, (B,), device=dev, generator=g)
for _ in range(30):
    tr.step(x, y)
torch.cuda.synchronize()
A.FUSED_TRACKING = False      # the launch-shape sweep is of the plain update kernel (the fused one has one shape)
R.graphed.reset()
shapes = [None, (0, 1, 0), (0, 2, 0), (0, 4, 0), (0, 1, 1), (0, 2, 1), (16, 1, 0), (32, 1, 0), (64, 2, 0), (32, 4, 0), None]
for rep in range(2):
    for sh in shapes:
        A.K1_SHAPE = sh
        for _ in range(3):
            tr.step(x, y)
        torch.cuda.synchronize()
        A.PROFILE_EVENTS = []
        for _ in range(12):
            tr.step(x, y)
        torch.cuda.synchronize()
        ev = A.PROFILE_EVENTS
        A.PROFILE_EVENTS = None
        gen = sorted(a.elapsed_time(b) * 1e3 for (n, i, a, b, gb, fl) in ev if i > 0)
        first = sorted(a.elapsed_time(b) * 1e3 for (n, i, a, b, gb, fl) in ev if i == 0)
        print(f"rep {rep} shape {sh}: general median {gen[len(gen) // 2]:.1f} us (min {gen[0]:.1f}), first-iteration {first[len(first) // 2]:.1f} us", flush=True)
