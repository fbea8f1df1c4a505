#!/usr/bin/env python3
"""Known-answer vectors for the reference's DLR criteria (``/root/reference/autopgd_train_clean.py:99-111``):
``dlr_loss`` and ``dlr_loss_targeted`` values and their autograd gradients on seeded logits.

Runs only in the build container (imports the reference); writes tests/golden/loss_dlr_vectors.npz, which is data
(inputs + the reference's outputs) and is what travels.  Usage: python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import autopgd_train_clean as ref  # noqa: E402  (reference, read-only)


def main():
    torch.set_num_threads(1)
    out = {}
    for tag, (B, C, scale) in {"a": (16, 10, 3.0), "b": (8, 1000, 2.0), "c": (5, 4, 1.0)}.items():
        g = torch.Generator().manual_seed(1234 + B * C)
        z = (torch.randn(B, C, generator=g) * scale).requires_grad_()
        y = torch.randint(0, C, (B,), generator=g)
        y[0] = int(z[0].argmax())                                  # a correctly classified sample
        order = z.detach().argsort(dim=1, descending=True)
        yt = order[:, min(2, C - 1)].clone()                       # 3rd most likely class as the target
        yt[y == yt] = order[y == yt, 0]
        lt = ref.dlr_loss_targeted(z, y, yt)
        (gt,) = torch.autograd.grad(lt.sum(), z)
        l1 = ref.dlr_loss(z, y)
        (g1,) = torch.autograd.grad(l1.sum(), z)
        out.update({f"{tag}_z": z.detach().numpy(), f"{tag}_y": y.numpy(), f"{tag}_yt": yt.numpy(),
                    f"{tag}_dlr_t": lt.detach().numpy(), f"{tag}_dlr_t_grad": gt.numpy(),
                    f"{tag}_dlr": l1.detach().numpy(), f"{tag}_dlr_grad": g1.numpy()})
    path = os.path.join(HERE, "loss_dlr_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
