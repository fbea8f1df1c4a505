#!/usr/bin/env python3
"""Generate golden APGD trajectories from the REFERENCE implementation.

Runs only in the build container (needs ``/root/reference``); the resulting
``apgd_*.npz`` files are committed and are the only thing that travels.  Each
fixture holds inputs plus what the reference's own
``apgd_train`` (``/root/reference/autopgd_train_clean.py:123-371``) produced:

  x, y, eps, n_iter, norm, soft          inputs
  logits[K+1,B,C], grads[K,B,...]        what the model returned at each call
  losses[K+1,B]                          F.cross_entropy(logits, y, 'none') as the reference computed it
  x_adv_sha[K+1]                         sha256 of the iterate handed to the model at each call
  x_adv_fed[K+1,B,...]                   the iterates themselves (small-K cases only)
  x_best, acc, loss_best, x_best_adv     the reference's return tuple

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/apgd_*.npz)
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import autopgd_train_clean as ref  # noqa: E402  (reference, read-only)


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().float().numpy().tobytes()).hexdigest()


class Tap(torch.autograd.Function):
    """Identity whose backward records the gradient w.r.t. the model input."""

    @staticmethod
    def forward(ctx, x, rec):
        ctx.rec = rec
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.rec.grads.append(g.detach().clone())
        return g, None


class Recorder(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self.xs, self.logits, self.grads = [], [], []

    def forward(self, x):
        self.xs.append(x.detach().clone())
        out = self.model(Tap.apply(x, self) if x.requires_grad else x)
        self.logits.append(out.detach().clone())
        return out


class ToyConv(nn.Module):
    def __init__(self, n_cls=10, width=8):
        super().__init__()
        self.c1 = nn.Conv2d(3, width, 3, padding=1)
        self.c2 = nn.Conv2d(width, width, 3, stride=2, padding=1)
        self.fc = nn.Linear(width, n_cls)
        for p in self.parameters():
            nn.init.normal_(p, std=0.6)

    def forward(self, x):
        x = F.gelu(self.c1(x))
        x = F.gelu(self.c2(x))
        return self.fc(x.mean((-2, -1))) * 3.0


class ToyMLP(nn.Module):
    def __init__(self, n_in, n_cls=7):
        super().__init__()
        self.f1 = nn.Linear(n_in, 16)
        self.f2 = nn.Linear(16, n_cls)
        for p in self.parameters():
            nn.init.normal_(p, std=0.5)

    def forward(self, x):
        return self.f2(torch.tanh(self.f1(x))) * 2.0


class _Scripted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, logits, grad):
        ctx.grad = grad
        return logits.clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.grad.clone(), None, None


class ScriptedModel(nn.Module):
    """Returns prescribed logits / input-gradients regardless of the input:
    lets a fixture put exact zeros, denormals and huge values into ``grad``."""

    def __init__(self, logits_seq, grad_seq):
        super().__init__()
        self.logits_seq, self.grad_seq, self.n = logits_seq, grad_seq, 0

    def forward(self, x):
        n = self.n
        self.n += 1
        if x.requires_grad:
            return _Scripted.apply(x, self.logits_seq[n], self.grad_seq[n])
        return self.logits_seq[n].clone()


def run_case(name, model, x, y, norm, eps, n_iter, soft=False, keep_fed=False, loss='ce'):
    if ONLY and name not in ONLY:
        return
    model.eval()
    rec = Recorder(model).eval()
    mixup = object() if soft else None
    xb, acc, lb, xba = ref.apgd_train(rec, x, y, norm=norm, eps=eps, n_iter=n_iter, mixup=mixup, loss=loss)
    assert len(rec.logits) == n_iter + 1 and len(rec.grads) == max(n_iter, 1), (len(rec.logits), len(rec.grads))   # K = 0 still runs fwd/bwd #0
    logits = torch.stack(rec.logits)
    losses = torch.stack([ref.criterion_dict[loss](l, y) for l in rec.logits])   # the reference's own criterion
    out = dict(
        x=x.numpy(), y=y.numpy(), eps=np.float64(eps), n_iter=np.int64(n_iter), norm=np.array(norm),
        soft=np.bool_(soft), loss=np.array(loss), channels_last=np.bool_(x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                                                    and not x.is_contiguous()),
        logits=logits.numpy(), grads=torch.stack(rec.grads).numpy(), losses=losses.numpy(),
        x_adv_sha=np.array([sha(t) for t in rec.xs]),
        x_best=xb.numpy(), acc=acc.numpy(), loss_best=lb.numpy(), x_best_adv=xba.numpy(),
    )
    if keep_fed:
        out["x_adv_fed"] = torch.stack(rec.xs).numpy()
    path = os.path.join(HERE, f"apgd_{name}.npz")
    np.savez_compressed(path, **out)
    moved = int((xb != x.clamp(0, 1)).flatten(1).any(1).sum())
    print(f"{name:22s} B={x.shape[0]} K={n_iter:3d} acc={acc.float().mean():.2f} "
          f"x_best moved for {moved}/{x.shape[0]} samples  {os.path.getsize(path) / 1024:.0f} KiB")


def labels_for(model, x, n_cls, g):
    with torch.no_grad():
        clean = model(x).argmax(1)
    rnd = torch.randint(0, n_cls, clean.shape, generator=g)
    y = clean.clone()
    y[1::2] = rnd[1::2]  # even samples start correctly classified, odd ones get a random label
    return y


ONLY = set(sys.argv[1:])      # optional: regenerate only the named cases


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)

    # --- Linf, hard labels, conv toy model -------------------------------------------------
    for k in (1, 2, 3, 5, 10):
        g = torch.Generator().manual_seed(100 + k)
        torch.manual_seed(k)
        m = ToyConv()
        x = torch.rand(6, 3, 12, 12, generator=g)
        y = labels_for(m, x, 10, g)
        run_case(f"linf_k{k}", m, x, y, "Linf", 4 / 255, k, keep_fed=(k <= 3))
    # --- n_iter = 0: `range(0)` (:209) - one forward/backward, the clamped clean point comes back (inputs partly outside [0,1])
    g = torch.Generator().manual_seed(99)
    torch.manual_seed(99)
    m = ToyConv()
    x = torch.rand(5, 3, 12, 12, generator=g) * 1.2 - 0.1
    y = labels_for(m, x.clamp(0, 1), 10, g)
    run_case("linf_k0", m, x, y, "Linf", 4 / 255, 0, keep_fed=True)
    for k, eps in ((25, 8 / 255), (100, 8 / 255)):
        g = torch.Generator().manual_seed(200 + k)
        torch.manual_seed(50 + k)
        m = ToyConv()
        x = torch.rand(4, 3, 8, 8, generator=g)
        y = labels_for(m, x, 10, g)
        run_case(f"linf_k{k}", m, x, y, "Linf", eps, k)

    # --- Linf, soft (mixup) labels ---------------------------------------------------------
    for k in (2, 10):
        g = torch.Generator().manual_seed(300 + k)
        torch.manual_seed(70 + k)
        m = ToyConv()
        x = torch.rand(6, 3, 12, 12, generator=g)
        y = torch.softmax(torch.randn(6, 10, generator=g) * 2.0, dim=1)
        run_case(f"linf_soft_k{k}", m, x, y, "Linf", 4 / 255, k, soft=True, keep_fed=(k <= 3))

    # --- Linf, channels_last input (memory format must be preserved) ------------------------
    g = torch.Generator().manual_seed(400)
    torch.manual_seed(90)
    m = ToyConv()
    x = torch.rand(5, 3, 10, 10, generator=g).contiguous(memory_format=torch.channels_last)
    y = labels_for(m, x, 10, g)
    run_case("linf_cl_k3", m, x, y, "Linf", 4 / 255, 3, keep_fed=True)

    # --- Linf, flat [B, F] input ------------------------------------------------------------
    g = torch.Generator().manual_seed(500)
    torch.manual_seed(91)
    m = ToyMLP(50)
    x = torch.rand(7, 50, generator=g)
    y = labels_for(m, x, 7, g)
    run_case("linf_flat_k5", m, x, y, "Linf", 8 / 255, 5, keep_fed=False)

    # --- Linf, scripted corner cases: exact 0/1 pixels, out-of-range x, zero / denormal / huge grads
    g = torch.Generator().manual_seed(600)
    B, K, C = 6, 4, 5
    x = torch.rand(B, 3, 6, 6, generator=g)
    x[0, :, :2] = 0.0
    x[1, :, :2] = 1.0
    x[2, 0] = -0.05          # below the box: x_adv is clamped, the eps-ball stays centred on x (:141, :222)
    x[3, 0] = 1.03
    x[4, 1, 0, :3] = torch.tensor([4 / 255, 1 - 4 / 255, 2 / 255])
    grads = torch.randn(K, B, 3, 6, 6, generator=g)
    grads[:, :, :, 0, 0] = 0.0
    grads[:, :, :, 0, 1] = -0.0
    grads[:, :, :, 0, 2] = 1e-42      # denormal
    grads[:, :, :, 0, 3] = -1e-42
    grads[:, :, :, 0, 4] = 3e38
    grads[1:, :, :, 0, 5] = float("nan")   # sign(nan) == 0
    logits = torch.randn(K + 1, B, C, generator=g) * 2
    logits[2] = logits[1]                 # equal losses: strict '>' must not fire (:321)
    y = torch.randint(0, C, (B,), generator=g)
    run_case("linf_scripted_k4", ScriptedModel(list(logits), list(grads)), x, y, "Linf", 4 / 255, K, keep_fed=True)

    # --- Linf with the DLR loss (criterion_dict['dlr'], :99-104) -------------------------------------
    g = torch.Generator().manual_seed(800)
    torch.manual_seed(95)
    m = ToyConv()
    x = torch.rand(6, 3, 12, 12, generator=g)
    y = labels_for(m, x, 10, g)
    run_case("linf_dlr_k5", m, x, y, "Linf", 8 / 255, 5, keep_fed=True, loss='dlr')

    # --- L2 ---------------------------------------------------------------------------------
    for k in (2, 10):
        g = torch.Generator().manual_seed(700 + k)
        torch.manual_seed(30 + k)
        m = ToyConv()
        x = torch.rand(6, 3, 12, 12, generator=g)
        y = labels_for(m, x, 10, g)
        run_case(f"l2_k{k}", m, x, y, "L2", 0.5, k, keep_fed=(k <= 3))


if __name__ == "__main__":
    main()
