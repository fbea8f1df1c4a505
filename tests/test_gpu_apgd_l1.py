"""The L1 branch of apgd_train on the device (revisiting_at_amd/apgd_l1.py) against the trajectories recorded from the reference
(tests/golden/apgd_l1_*.npz) and against the numpy oracle.  Agreement is to fp32 summation-order noise (the projection sums and
prefix-sums thousands of terms in a different order on the GPU), like the L2 branch; `acc` and the invariants are exact."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT  # noqa: F401
from oracle import apgd_oracle as O
from oracle import apgd_l1_oracle as L1

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "apgd_l1_*.npz")))


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R_
    R_._lib.load()
    return R_


class _Scripted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, logits, grad):
        ctx.grad = grad
        return logits.clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.grad.clone(), None, None


class Replay(torch.nn.Module):
    """Call n returns the reference model's recorded logits[n] (and gradient), whatever the input; keeps what it was fed."""

    def __init__(self, logits, grads):
        super().__init__()
        self.logits, self.grads, self.n, self.fed = logits, grads, 0, []

    def forward(self, x):
        n = self.n
        self.n += 1
        self.fed.append(x.detach().clone())
        if x.requires_grad:
            return _Scripted.apply(x, self.logits[n], self.grads[n])
        return self.logits[n].clone()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_l1_attack_follows_the_reference_trajectory(R, path):
    f = np.load(path)
    K = int(f["n_iter"])
    m = Replay(torch.from_numpy(f["logits"]).cuda(), torch.from_numpy(f["grads"]).cuda()).eval()
    x, y = torch.from_numpy(f["x"]).cuda(), torch.from_numpy(f["y"]).cuda()
    xb, acc, lb, xba = R.apgd_train(m, x, y, norm="L1", eps=float(f["eps"]), n_iter=K, mixup=object() if bool(f["soft"]) else None,
                                    is_train=bool(f["is_train"]))
    fed = torch.stack(m.fed).cpu().numpy()
    assert fed.shape == f["x_adv_fed"].shape
    err = np.abs(fed - f["x_adv_fed"]).reshape(K + 1, -1).max(1)
    assert float(err.max()) <= 5e-6, err
    assert np.array_equal(acc.cpu().numpy(), f["acc"])
    np.testing.assert_allclose(lb.cpu().numpy(), f["loss_best"], rtol=1e-6, atol=1e-7)
    assert float(np.abs(xb.cpu().numpy() - f["x_best"]).max()) <= 5e-6
    assert float(np.abs(xba.cpu().numpy() - f["x_best_adv"]).max()) <= 5e-6
    assert xb.shape == x.shape and not xb.requires_grad


def test_l1_projection_vs_oracle_and_invariants_at_image_size(R):
    from revisiting_at_amd import apgd_l1
    g = torch.Generator(device="cuda").manual_seed(0)
    B, n = 8, 3 * 64 * 64
    x = torch.rand(B, n, device="cuda", generator=g)
    y = torch.randn(B, n, device="cuda", generator=g) * 0.2 * (torch.rand(B, n, device="cuda", generator=g) < 0.1)
    for eps in (2.0, 12.0, 1e4):
        d = apgd_l1.l1_projection(x, y, eps)
        want = L1.l1_projection(x.cpu().numpy(), y.cpu().numpy(), eps)
        assert float(np.abs(d.cpu().numpy() - want).max()) <= 2e-5
        z = x + y + d
        assert float(z.min()) >= -1e-6 and float(z.max()) <= 1 + 1e-6
        assert float((y + d).abs().sum(1).max()) <= eps * (1 + 1e-5) + 1e-4


def test_l1_attack_on_a_live_model_matches_the_oracle_and_stays_in_the_ball(R):
    """A small conv net evaluated on the device for the product and on the CPU (same weights) for the oracle: the attack only sees
    |grad| ranks and signs, so the two trajectories agree except where bf16-free fp32 noise reorders ties - statistically equal."""
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.GELU(), torch.nn.Conv2d(8, 8, 3, stride=2, padding=1),
                              torch.nn.GELU(), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(8, 10)).eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(6, 3, 16, 16, generator=g)
    with torch.no_grad():
        y = net(x).argmax(1)
    eps, K = 8.0, 10
    oxb, oacc, olb, oxba, _ = L1.apgd_train_l1_oracle(O.TorchModelAdapter(net, y.numpy()), x.numpy(), y.numpy(), eps, K)
    dev = net.cuda()
    xb, acc, lb, xba = R.apgd_train(dev, x.cuda(), y.cuda(), norm="L1", eps=eps, n_iter=K)
    d = (xb.cpu() - x).flatten(1)
    assert float(d.abs().sum(1).max()) <= eps * (1 + 1e-5) and float(xb.min()) >= 0 and float(xb.max()) <= 1
    assert float((xb.cpu().numpy() == oxb).mean()) >= 0.97 and float(np.abs(xb.cpu().numpy() - oxb).max()) <= 1.0
    np.testing.assert_allclose(lb.cpu().numpy(), olb, rtol=2e-2, atol=1e-3)
    assert np.array_equal(acc.cpu().numpy(), oacc)
    # and through the product ConvNeXt under bf16 autocast: invariants only
    A = R.architecture
    cn = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), num_classes=10)
    cn.stem = A.ConvBlock1(48)
    cn = cn.cuda().to(memory_format=torch.channels_last).eval()
    xc = torch.rand(4, 3, 64, 64, device="cuda")
    yc = torch.randint(0, 10, (4,), device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xb2, acc2, lb2, _ = R.apgd_train(cn, xc, yc, norm="L1", eps=12.0, n_iter=5)
    d2 = (xb2 - xc).flatten(1)
    assert float(d2.abs().sum(1).max()) <= 12.0 * (1 + 1e-5) and float(xb2.min()) >= 0 and float(xb2.max()) <= 1
    assert bool(torch.isfinite(lb2).all())
