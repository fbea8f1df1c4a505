#!/usr/bin/env python3
"""BASELINE config #5 in isolation for a kernel profile: APGD-CE (evaluation attack, fp32 as AA_eval.py runs it) on ConvNeXt-B-CvSt,
batch 32, a few iterations.  Usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/probe/cfg5_profile.py [n_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
dev = torch.device("cuda")
model = R.get_new_model("convnext_base", pretrained=False, not_original=True).to(dev).to(memory_format=torch.channels_last).eval()
for p in model.parameters():
    p.requires_grad_(False)
g = torch.Generator(device=dev).manual_seed(9)
x = torch.rand(32, 3, 224, 224, device=dev, generator=g)
with torch.no_grad():
    y = model(x).argmax(1)                 # every point starts robust: the attack runs on the whole batch (as bench.py does)
R.aa_eval.apgd_attack(model, x, y, "Linf", 4 / 255, 3, "ce", None, True, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
R.aa_eval.apgd_attack(model, x, y, "Linf", 4 / 255, n_iter, "ce", None, True, g)
torch.cuda.synchronize()
print("s per iteration", (time.perf_counter() - t0) / n_iter)
