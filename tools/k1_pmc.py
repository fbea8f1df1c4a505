#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the APGD Linf update at BASELINE config #2 size
(B=256, 3x224x224, fp32) plus a plain device copy of known size as the byte-count calibration
(MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R

lib = R._lib.load()
B, E, eps = 256, 3 * 224 * 224, 4 / 255
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(B, E, device="cuda", generator=g)
xa = (x + (torch.rand(B, E, device="cuda", generator=g) * 2 - 1) * eps).clamp(0, 1)
xo = (x + (torch.rand(B, E, device="cuda", generator=g) * 2 - 1) * eps).clamp(0, 1)
gr = torch.randn(B, E, device="cuda", generator=g) * 1e-3
step = torch.full((B,), 2 * eps, device="cuda")
out = torch.empty_like(x)
S = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), gr.data_ptr(), 0, step.data_ptr(),
                                  out.data_ptr(), None, B, E, eps, 0.75, S) == 0
    out.copy_(x)          # calibration: 154.1 MB read + 154.1 MB written
for _ in range(10):       # iteration-0 form: x_adv_old aliases x_adv (a = 1.0), 16 algorithmic B/elem
    assert lib.apgd_linf_step_f32_ex(x.data_ptr(), xa.data_ptr(), xa.data_ptr(), gr.data_ptr(), 0, step.data_ptr(),
                                     out.data_ptr(), None, B, E, eps, 1.0, 0, 2, 0, S) == 0
torch.cuda.synchronize()
print("done", B * E * 4)
