#!/usr/bin/env python3
"""Golden APGD-L1 trajectories from the REFERENCE (``/root/reference/autopgd_train_clean.py:160-168, 239-250, 351-362`` and
``L1_projection`` ``:24-91``): sparse signed steps with an adaptive top-k, projection onto the intersection of the L1 ball and the
[0, 1] box.  Same layout as ``apgd_*.npz`` (see make_golden.py); build container only.

Usage: python tests/golden/make_l1_golden.py   (rewrites tests/golden/apgd_l1_*.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference)

ref = G.ref


def run_case(name, model, x, y, eps, n_iter, is_train=True, soft=False):
    model.eval()
    rec = G.Recorder(model).eval()
    xb, acc, lb, xba = ref.apgd_train(rec, x, y, norm="L1", eps=eps, n_iter=n_iter, mixup=object() if soft else None, is_train=is_train)
    logits = torch.stack(rec.logits)
    losses = torch.stack([ref.criterion_dict["ce"](l, y) for l in rec.logits])
    out = dict(x=x.numpy(), y=y.numpy(), eps=np.float64(eps), n_iter=np.int64(n_iter), norm=np.array("L1"), soft=np.bool_(soft),
               is_train=np.bool_(is_train), logits=logits.numpy(), grads=torch.stack(rec.grads).numpy(), losses=losses.numpy(),
               x_adv_fed=torch.stack(rec.xs).numpy(), x_best=xb.numpy(), acc=acc.numpy(), loss_best=lb.numpy(), x_best_adv=xba.numpy())
    path = os.path.join(HERE, f"apgd_l1_{name}.npz")
    np.savez_compressed(path, **out)
    d = (xb - x.clamp(0, 1)).flatten(1)
    print(f"{name:14s} B={x.shape[0]} K={n_iter:3d} acc={acc.float().mean():.2f} max L1={float(d.abs().sum(1).max()):.4f} (eps {eps}) "
          f"nonzero/sample={float((d != 0).sum(1).float().mean()):.0f}  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    for k, eps, is_train in ((1, 6.0, True), (3, 6.0, True), (10, 12.0, True), (25, 12.0, True), (10, 12.0, False)):
        g = torch.Generator().manual_seed(300 + k)
        torch.manual_seed(k)
        m = G.ToyConv()
        x = torch.rand(5, 3, 12, 12, generator=g)
        y = G.labels_for(m, x, 10, g)
        run_case(f"k{k}" + ("" if is_train else "_eval"), m, x, y, eps, k, is_train=is_train)
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(4)
    m = G.ToyConv()
    # (a channels-last x makes the reference itself fail at `grad.abs().view(B, -1)`, :240 - its L1 branch takes NCHW-contiguous
    #  inputs only; this case has inputs partly outside [0, 1] instead)
    x = torch.rand(4, 3, 8, 8, generator=g) * 1.2 - 0.1
    run_case("outside_k5", m, x, G.labels_for(m, x, 10, g), 4.0, 5)
    ys = torch.softmax(torch.randn(5, 10, generator=g) * 2, 1)
    x = torch.rand(5, 3, 12, 12, generator=g)
    run_case("soft_k10", m, x, ys, 12.0, 10, soft=True)


if __name__ == "__main__":
    main()
