#!/usr/bin/env python3
"""Per-component hip-vs-library comparison at given shapes (bf16 autocast): one ConvNeXt block per stage shape, the
downsample layers and the ConvStem, forward / input gradient (attack mode and training mode) / parameter gradients.
Usage: python tools/debug_blocks.py [arch] [res] [batch]   (default convnext_large 320 2)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import revisiting_at_amd as R

arch = sys.argv[1] if len(sys.argv) > 1 else "convnext_large"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 320
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(mod, x, mode, attack):
    R.ops.MODE = mode
    R.ops.invalidate_weight_cache()
    torch.clear_autocast_cache()
    for p in mod.parameters():
        p.grad = None
    xi = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = mod(xi)
    cot = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(out.dtype)
    if attack:
        with R.ops.input_grad_only():
            (gx,) = torch.autograd.grad([out], [xi], grad_outputs=[cot])
        return out.detach(), gx, {}
    out.backward(cot)
    return out.detach(), xi.grad, {n: p.grad.detach().clone() for n, p in mod.named_parameters()}


torch.manual_seed(0)
model = R.get_new_model(arch, pretrained=False, not_original=True, img_size=res).cuda().to(memory_format=torch.channels_last)
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.endswith("gamma"):
            p.fill_(0.5)
model.eval()
x = torch.rand(nb, 3, res, res, device="cuda")
# capture the inputs of every component with hooks
inputs = {}


def hook(name):
    def f(m, inp):
        inputs[name] = inp[0].detach()
    return f


comps = {}
if hasattr(model, "stem"):
    comps["stem"] = model.stem
if hasattr(model, "stages"):
    for i, st in enumerate(model.stages):
        if not isinstance(st.downsample, torch.nn.Identity):
            comps[f"stage{i}.downsample"] = st.downsample
        comps[f"stage{i}.block0"] = st.blocks[0]
        comps[f"stage{i}.block1"] = st.blocks[1]
    comps["head"] = model.head
hs = [m.register_forward_pre_hook(hook(n)) for n, m in comps.items()]
R.ops.MODE = "eager"
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    model(x)
for h in hs:
    h.remove()

print(f"{arch} {res}x{res} batch {nb}: relative L2 error hip vs library composition (bf16 autocast)")
for name, mod in comps.items():
    xin = inputs[name]
    for attack in (True, False):
        oh, gh, ph = run(mod, xin, "hip", attack)
        oe, ge, pe = run(mod, xin, "eager", attack)
        worst = max(((rel(ph[k], pe[k]), k) for k in pe if float(pe[k].float().norm()) > 0), default=(0.0, "-"))
        print(f"{name:20s} in {tuple(xin.shape)} {str(xin.dtype)[6:]:9s} {'attack' if attack else 'train '}  fwd {rel(oh, oe):.2e}  "
              f"dgrad {rel(gh, ge):.2e}  worst param grad {worst[0]:.2e} ({worst[1]})")
