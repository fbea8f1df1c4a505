#!/usr/bin/env python3
"""Is the whole AT step reproducible run to run?  Two trainers from the same seed, K steps each on the same batches: are the
parameters bit-identical afterwards (fixed-order partial sums everywhere: cnx_gemm_tn, the depthwise / LayerNorm partials, the
gradient identities)?  Usage: python tools/probe/train_determinism.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")


def run():
    torch.manual_seed(0)
    model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True, img_size=224)
    tr = R.ATTrainStep(model, "convnext_tiny", R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2, graph=1), dev, lr=1e-3,
                       amp_dtype=torch.bfloat16, ema=True)
    g = torch.Generator(device=dev).manual_seed(7)
    losses = []
    for _ in range(steps):
        x = torch.rand(batch, 3, 224, 224, device=dev, generator=g)
        y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
        losses.append(float(tr.step(x, y)))
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in model.parameters()]


la, pa = run()
lb, pb = run()
n_diff = sum(int(not torch.equal(a, b)) for a, b in zip(pa, pb))
worst = max(float((a - b).abs().max()) for a, b in zip(pa, pb))
print("losses", la, lb)
print(f"parameters differing between the two runs: {n_diff} of {len(pa)} (largest |difference| {worst:.3e})")
