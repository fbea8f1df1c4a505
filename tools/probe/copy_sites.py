#!/usr/bin/env python3
"""Where the ATen copy / fill kernels of one eager AT step come from: torch.profiler with stacks, device time by (op, shape, first repo frame)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
torch.manual_seed(0)
R.ops.load_gemm_table()
model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True, img_size=224)
tr = R.ATTrainStep(model, "convnext_tiny", R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2), dev, lr=1e-3, amp_dtype=torch.bfloat16, ema=True)   # eager: adv.graph unset
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand(256, 3, 224, 224, device=dev, generator=g)
y = torch.randint(0, 1000, (256,), device=dev, generator=g)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    sdt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if ev.key.startswith("aten::") and sdt > 0:
        rows.append((sdt, ev.count, ev.key, str(ev.input_shapes)[:110]))
for sdt, c, k, sh in sorted(rows, reverse=True)[:45]:
    print(f"{sdt:9.1f} us {c:4d}  {k:28s} {sh}")
