#!/usr/bin/env python3
"""Is one forward + input-gradient backward of a product model bit-reproducible run to run?  (The attack's trajectory is a function
of sign(grad): one flipped bit in a tiny gradient moves a pixel by a whole step.)  Usage: python tools/determinism_check.py [arch] [res] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import revisiting_at_amd as R

arch = sys.argv[1] if len(sys.argv) > 1 else "convnext_iso"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
torch.manual_seed(0)
m = R.get_new_model(arch, pretrained=False, not_original=True, updated=False)
m = R.normalize_model(m, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)) if hasattr(R, "normalize_model") else m
m = m.cuda().eval().to(memory_format=torch.channels_last)
for p in m.parameters():
    p.requires_grad_(False)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(bs, 3, res, res, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 1000, (bs,), device="cuda", generator=g)
acts = {}


def hook(name):
    def f(mod, inp, out):
        if isinstance(out, torch.Tensor):
            acts.setdefault(name, []).append(out.detach().float().clone())
    return f


gacts = {}
order = []


def bhook(name):
    def f(mod, gin, gout):
        if gin and isinstance(gin[0], torch.Tensor):
            if name not in gacts:
                order.append(name)
            gacts.setdefault(name, []).append(gin[0].detach().float().clone())
    return f


for n, mod in m.named_modules():
    if n and n.count(".") <= 5:
        mod.register_forward_hook(hook(n))
        mod.register_full_backward_hook(bhook(n))


def once():
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lo = m(xi)
        loss = F.cross_entropy(lo.float(), y)
    (gr,) = torch.autograd.grad(loss, xi)
    return lo.detach().float().clone(), gr.detach().clone()


l0, g0 = once()
bad_l = bad_g = 0
for _ in range(reps):
    l1, g1 = once()
    bad_l += int(not torch.equal(l0, l1))
    bad_g += int(not torch.equal(g0, g1))
print(f"{arch}@{res} batch {bs}: logits differed in {bad_l}/{reps} repeats, input gradients in {bad_g}/{reps}")
first = None
for n, lst in acts.items():
    d = sum(int(not torch.equal(lst[0], t)) for t in lst[1:])
    if d and first is None:
        first = n
    if d:
        print(f"  forward output of {n}: differed in {d}/{len(lst) - 1}")
print("first differing forward module:", first)
firstb = None
for n in order:                                            # backward order: first entry is closest to the loss
    lst = gacts[n]
    d = sum(int(not torch.equal(lst[0], t)) for t in lst[1:])
    if d:
        if firstb is None:
            firstb = n
        k = next(i for i, t in enumerate(lst[1:]) if not torch.equal(lst[0], t))
        diff = (lst[0] - lst[1 + k]).abs()
        print(f"  grad_input of {n} {tuple(lst[0].shape)}: differed in {d}/{len(lst) - 1}; elements {int((diff > 0).sum())}, max |diff| {float(diff.max()):.3e}, max |g| {float(lst[0].abs().max()):.3e}")
print("first differing backward module (closest to the loss):", firstb)
