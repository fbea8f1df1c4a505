#!/usr/bin/env python3
"""A few adversarial-training steps (ConvNeXt-T-CvSt, APGD-2, bf16) through the hand-written path and through the plain library
composition (APGD_OPS=eager semantics) from the same initial weights and data: loss trajectories side by side."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R

def run(mode, steps=4, B=16):
    R.ops.MODE = mode
    torch.manual_seed(0)
    model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True)
    adv = R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2)
    tr = R.ATTrainStep(model, "convnext_tiny", adv, "cuda", lr=1e-3, ema=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    losses = []
    for i in range(steps):
        x = torch.rand(B, 3, 224, 224, device="cuda", generator=g)
        y = torch.randint(0, 1000, (B,), device="cuda", generator=g)
        losses.append(float(tr.step(x, y)))
    return losses

h = run("hip"); e = run("eager")
for i, (a, b) in enumerate(zip(h, e)):
    print(f"step {i}: hip {a:.4f}  eager {b:.4f}  diff {a - b:+.4f}")
