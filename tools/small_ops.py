#!/usr/bin/env python3
"""Where do the small kernels of the training pass come from?  One eager AT step of the benchmark configuration under torch.profiler
(CPU side, with Python stacks): aten::copy_ / fill_ / zero_ / _to_copy / sum ... calls grouped by the innermost frame inside the package.
usage: tools/small_ops.py [arch=convnext_tiny] [batch=256] [phase=train|attack|all]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
from torch.profiler import profile, ProfilerActivity
arch = sys.argv[1] if len(sys.argv) > 1 else "convnext_tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
phase = sys.argv[3] if len(sys.argv) > 3 else "train"
dev = torch.device("cuda")
torch.manual_seed(0)
model = R.get_new_model(arch, pretrained=False, not_original=True)
tr = R.ATTrainStep(model, arch, R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2, graph=0), dev, lr=1e-3, channels_last=True,
                   amp_dtype=torch.bfloat16, ema=True, gemm_table=True, graph_train=False)
x = torch.rand(B, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
z = tr._perturbed(x, y) if phase == "train" else None
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    if phase == "train":
        tr.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = tr.inner.base_model(z)
            loss = tr.loss(out, y)
        loss.backward()
        tr.optimizer.step()
        R.ops.invalidate_weight_cache()
        tr.ema.update()
    elif phase == "attack":
        tr._perturbed(x, y)
    else:
        tr.step(x, y)
    torch.cuda.synchronize()
ev = prof.events()
# kernels launched per CPU op: use the op's own device time / kernel count where available; group leaf aten ops
want = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::_to_copy", "aten::sum", "aten::mul", "aten::add", "aten::add_", "aten::mul_", "aten::clone",
        "aten::contiguous", "aten::empty_like", "aten::zeros", "aten::mean", "aten::div", "aten::cat", "aten::index_select")
groups = collections.Counter()
times = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or e.name not in want:
        continue
    kern = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kern:
        continue
    frame = "?"
    for f in (e.stack or []):
        if "revisiting" in f and "torch/" not in f:
            frame = f.split("revisiting-at_amd/")[-1].split("revisiting_at_amd/")[-1]
            break
    else:
        for f in (e.stack or []):
            if "torch/optim" in f or "autograd" in f:
                frame = f.split("site-packages/")[-1]
                break
    shp = str(e.input_shapes)[:60]
    groups[(e.name, frame[:90], shp)] += len(kern)
    times[(e.name, frame[:90], shp)] += sum(k.duration for k in kern)
print(f"{'kernels':>7s} {'us':>8s}  op / frame / shapes")
for k, c in sorted(groups.items(), key=lambda kv: -times[kv[0]])[:70]:
    print(f"{c:7d} {times[k]:8.1f}  {k[0]:16s} {k[1]:90s} {k[2]}")
print("total small-op kernels", sum(groups.values()), "us", round(sum(times.values()), 1))
# every kernel under 12 us, by name
kc = collections.Counter(); kt = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA and e.device_time_total < 12 and e.name and not e.name.startswith("Memcpy") or (e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" in e.name):
        kc[e.name[:100]] += 1; kt[e.name[:100]] += e.device_time_total
print("kernels under 12 us (and memcpys):")
for k, c in sorted(kc.items(), key=lambda kv: -kt[kv[0]])[:40]:
    print(f"{c:6d} {kt[k]:8.1f} us  {k}")
