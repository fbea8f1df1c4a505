#!/usr/bin/env python3
"""Golden vectors for the second value of the reference's attack selector, ``adv.attack=fgsm``
(``/root/reference/main.py:836-842`` -> ``/root/reference/fgsm_train.py:72-100``: random start, one signed gradient step, projection).

Runs only in the build container (needs ``/root/reference``).  ``fgsm_train.py`` imports ``robustbench`` and ``autoattack`` at module
level without using them in ``fgsm_train``; neither is installed here, so two EMPTY placeholder modules are registered before the
import (nothing of theirs is called).  The uniform draw ``t = torch.rand_like(x)`` (``:81``) is recorded by re-seeding the global
generator with the same seed right before the call.  Each fixture: inputs (x, y, eps, alpha, noise_level, flags, t), what the model
returned (logits, grad, the iterate it was fed) and the reference's output x_adv.

Usage: python tests/golden/make_fgsm_golden.py   (rewrites tests/golden/fgsm_*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
for name in ("robustbench", "autoattack"):
    sys.modules.setdefault(name, types.ModuleType(name))
import fgsm_train as ref  # noqa: E402  (reference, read-only)
from make_golden import Recorder, ToyConv, ToyMLP, labels_for  # noqa: E402


def run_case(name, model, x, y, eps, seed, **kw):
    model.eval()
    rec = Recorder(model).eval()
    torch.manual_seed(seed)
    t = torch.rand_like(x)
    torch.manual_seed(seed)
    out = ref.fgsm_train(rec, x, y, eps, **kw)
    assert len(rec.logits) == 1 and len(rec.grads) == 1
    d = dict(x=x.numpy(), y=y.numpy(), eps=np.float64(eps), t=t.numpy(), logits=rec.logits[0].numpy(), grad=rec.grads[0].numpy(),
             x_fed=rec.xs[0].numpy(), x_adv=out.detach().numpy(),
             alpha=np.float64(kw.get("alpha", 1.25)), noise_level=np.float64(kw.get("noise_level", 1.0)),
             use_rs=np.bool_(kw.get("use_rs", False)), skip_projection=np.bool_(kw.get("skip_projection", False)),
             soft=np.bool_(y.dtype.is_floating_point))
    path = os.path.join(HERE, f"fgsm_{name}.npz")
    np.savez_compressed(path, **d)
    print(f"{name:18s} B={x.shape[0]} max|x_adv-x|={float((out - x).abs().max()):.5f} range [{float(out.min()):.3f}, {float(out.max()):.3f}] "
          f"{os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(1)
    m = ToyConv()
    x = torch.rand(6, 3, 12, 12, generator=g)
    y = labels_for(m, x, 10, g)
    eps = 4 / 255
    # the trainer's call (main.py:836-842): use_rs=True, alpha / noise_level / skip_projection from the flags
    run_case("rs_default", m, x, y, eps, 11, use_rs=True, alpha=1.0, noise_level=1.0, skip_projection=False)
    run_case("rs_nfgsm", m, x, y, eps, 12, use_rs=True, alpha=1.25, noise_level=2.0, skip_projection=True)
    run_case("rs_alpha2", m, x, y, 8 / 255, 13, use_rs=True, alpha=2.0, noise_level=0.5, skip_projection=False)
    run_case("plain", m, x, y, eps, 14)                                       # the function's own defaults: no random start
    run_case("plain_skip", m, x, y, eps, 15, skip_projection=True, alpha=1.0)
    # channels-last input, inputs partly outside [0, 1], soft labels, a flat (2-D) input
    xc = (torch.rand(4, 3, 8, 8, generator=g) * 1.4 - 0.2).contiguous(memory_format=torch.channels_last)
    run_case("rs_cl_outside", m, xc, labels_for(m, xc, 10, g), eps, 16, use_rs=True, alpha=1.25, noise_level=1.0)
    ys = torch.softmax(torch.randn(6, 10, generator=g), 1)
    run_case("rs_soft", m, x, ys, eps, 17, use_rs=True, alpha=1.25, noise_level=1.0)
    torch.manual_seed(2)
    mm = ToyMLP(20)
    xf = torch.rand(5, 20, generator=g)
    run_case("rs_flat", mm, xf, labels_for(mm, xf, 7, g), 0.05, 18, use_rs=True, alpha=1.25, noise_level=1.0)


if __name__ == "__main__":
    main()
