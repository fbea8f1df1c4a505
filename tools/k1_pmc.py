#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the APGD Linf update at BASELINE config #2 size (B=256, 3x224x224) in its four
forms - general (i > 0) / first iteration (i = 0, x_adv_old aliases x_adv), each with an fp32 gradient and with int8
gradient signs - plus a plain device copy of known size as the byte-count calibration (MI355X_MICROARCH.md §HBM:
FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950).  tools/k1_traffic.py turns the two passes into
profiles/k1_traffic.json."""
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
sg = torch.sign(gr).to(torch.int8)
step = torch.full((B,), 2 * eps, device="cuda")
out = torch.empty_like(x)
S = torch.cuda.current_stream().cuda_stream
sb = R.ops.signs_to_blocked(sg)                      # the order the product's stem kernel writes (APGD_I8_BLK)
for grad, code in ((gr, 0), (sg, 3), (sb, 4)):
    for _ in range(10):
        assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), grad.data_ptr(), code, step.data_ptr(),
                                      out.data_ptr(), None, B, E, eps, 0.75, S) == 0
        out.copy_(x)          # calibration: 154.1 MB read + 154.1 MB written
    for _ in range(10):       # iteration-0 form: x_adv_old aliases x_adv (a = 1.0)
        assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xa.data_ptr(), grad.data_ptr(), code, step.data_ptr(),
                                      out.data_ptr(), None, B, E, eps, 1.0, S) == 0
# round 5: the fused form (apgd_linf_step_track_f32: the step + the row moves of the iteration before it).  Per gradient type, in this
# order (tools/k1_traffic.py slices the dispatches of a kernel name by it): 10 launches with every sample NEW_BEST | MISCLS (what a
# random-init model's APGD-2 step looks like), 10 with every sample NEW_BEST only, then 10 of the iteration-0 form (flags == NULL)
xb, xba = torch.empty_like(x), torch.empty_like(x)
for grad, code in ((gr, 0), (sg, 3)):
    gb = torch.empty_like(grad)
    for fv in (3, 1):
        flags = torch.full((B,), fv, device="cuda", dtype=torch.uint8)
        for _ in range(10):
            assert lib.apgd_linf_step_track_f32(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), grad.data_ptr(), code, step.data_ptr(),
                                                out.data_ptr(), flags.data_ptr(), xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), B, E,
                                                eps, 0.75, S) == 0
    for _ in range(10):
        assert lib.apgd_linf_step_track_f32(x.data_ptr(), xa.data_ptr(), xa.data_ptr(), grad.data_ptr(), code, step.data_ptr(),
                                            out.data_ptr(), None, xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), B, E, eps, 1.0, S) == 0
torch.cuda.synchronize()
print("done", B * E * 4)
