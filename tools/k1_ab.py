#!/usr/bin/env python3
"""K1 at BASELINE config #2 size (B=256, 3x224x224), its gradient forms timed alternately in one process: fp32, int8 signs in
element order (round 2), int8 signs in the blocked order (round 3), general and first-iteration launch; cold-MALL variant (a 512 MB
copy between launches evicts the tensors, as the model's kernels do inside the step)."""
import os, sys
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
sb = R.ops.signs_to_blocked(sg)
step = torch.full((B,), 2 * eps, device="cuda")
out = torch.empty_like(x)
junk_a, junk_b = torch.empty(128 * 1024 * 1024, device="cuda"), torch.empty(128 * 1024 * 1024, device="cuda")
S = torch.cuda.current_stream().cuda_stream
forms = {"fp32": (gr, 0), "int8": (sg, 3), "int8 blocked": (sb, 4)}
N = B * E
for cold in (False, True):
    for first in (False, True):
        ts = {k: [] for k in forms}
        for r in range(24):
            for k, (gt, code) in forms.items():
                if cold:
                    junk_b.copy_(junk_a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xa.data_ptr() if first else xo.data_ptr(), gt.data_ptr(), code,
                                            step.data_ptr(), out.data_ptr(), None, B, E, eps, 1.0 if first else 0.75, S)
                e1.record(); e1.synchronize()
                assert rc == 0
                if r >= 4:
                    ts[k].append(e0.elapsed_time(e1) * 1e3)
        alg = (16 if first else 20) * N
        for k, v in ts.items():
            v.sort()
            med = v[len(v) // 2]
            moved = alg if k == "fp32" else alg - 3 * N
            print(f"{'cold' if cold else 'warm'} {'first' if first else 'general':8s} {k:13s} {med:7.1f} us (min {v[0]:7.1f})  algorithmic {alg / med / 1e6:6.2f} TB/s = "
                  f"{alg / med / 1e6 / 8:.3f}   moved {moved / med / 1e6:6.2f} TB/s = {moved / med / 1e6 / 8:.3f}", flush=True)
