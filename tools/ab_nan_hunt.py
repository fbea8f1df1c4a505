#!/usr/bin/env python3
"""Hunt a non-finite loss of the 'round4' kernel set (bench.py extra.ab, gpurun_out/r6a): fresh trainers under the set, per-step
checks of loss / gradients / parameters, first offender named.  usage: tools/ab_nan_hunt.py [set] [reps] [steps] [graph] [override-dict] [arch] [res] [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import revisiting_at_amd as R  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "round4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 1
override = eval(sys.argv[5]) if len(sys.argv) > 5 else {}
arch = sys.argv[6] if len(sys.argv) > 6 else "convnext_tiny"
res = int(sys.argv[7]) if len(sys.argv) > 7 else 224
batch = int(sys.argv[8]) if len(sys.argv) > 8 else 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.rand(batch, 3, res, res, device=dev, generator=g)
y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
for rep in range(reps):
    R.ops.kernel_set(name)
    if override:
        R.ops.kernel_set(override)
    R.graphed.reset()
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    model = R.get_new_model(arch, pretrained=False, not_original=True, img_size=res)
    tr = R.ATTrainStep(model, arch, R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2, graph=graph), dev, lr=1e-3,
                       channels_last=True, amp_dtype=torch.bfloat16, ema=True, gemm_table=True, graph_train=bool(graph))
    bad_at = None
    losses = []
    for i in range(steps):
        loss = tr.step(x, y)
        torch.cuda.synchronize()
        losses.append(round(float(loss), 4))
        badp = [n for n, p in tr.inner.named_parameters() if not torch.isfinite(p).all()]
        badg = [n for n, p in tr.inner.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        if badp or badg or not torch.isfinite(loss):
            bad_at = i
            print(f"rep {rep} step {i}: loss {float(loss)} bad params {badp[:8]} ({len(badp)}) bad grads {badg[:8]} ({len(badg)})", flush=True)
            break
    print(f"rep {rep} {arch}@{res} b{batch} set {name} {override} graph {graph}: losses {losses} bad_at {bad_at}", flush=True)
    del tr, model
