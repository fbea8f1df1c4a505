"""adv.attack=fgsm on the device (revisiting_at_amd.fgsm_train -> apgd_fgsm_start_f32 / apgd_fgsm_step_f32) against the vectors
recorded from the reference's ``fgsm_train`` (tests/golden/fgsm_*.npz) and against the numpy oracle: bit for bit."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT  # noqa: F401
from oracle import apgd_oracle as O
from oracle import fgsm_oracle as FO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "fgsm_*.npz")))


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
        # the attack differentiates sum CE: hand back the RECORDED input gradient whatever dlogits arrives
        return ctx.grad.clone(), None, None


class Scripted(torch.nn.Module):
    """Returns the logits / input gradient the reference's model returned, and keeps the iterate it was fed."""

    def __init__(self, logits, grad):
        super().__init__()
        self.logits, self.grad, self.fed = logits, grad, None

    def forward(self, x):
        self.fed = x.detach().clone()
        return _Scripted.apply(x, self.logits, self.grad)


def _fmt(x, like_cl):
    t = torch.from_numpy(x).cuda()
    return t.contiguous(memory_format=torch.channels_last) if like_cl and t.dim() == 4 else t


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_fgsm_train_reproduces_the_reference_vectors_bit_for_bit(R, path):
    f = np.load(path)
    cl = "cl_" in path
    x, t = _fmt(f["x"], cl), _fmt(f["t"], cl)
    y = torch.from_numpy(f["y"]).cuda()
    m = Scripted(torch.from_numpy(f["logits"]).cuda(), _fmt(f["grad"], cl)).eval()
    out = R.fgsm_train(m, x, y, float(f["eps"]), alpha=float(f["alpha"]), use_rs=bool(f["use_rs"]), noise_level=float(f["noise_level"]),
                       skip_projection=bool(f["skip_projection"]), _t=t)
    assert np.array_equal(m.fed.cpu().numpy(), f["x_fed"]), "start point"
    assert np.array_equal(out.cpu().numpy(), f["x_adv"])
    assert out.shape == x.shape and out.stride() == x.stride() and not out.requires_grad


@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16, torch.int8])
@pytest.mark.parametrize("project", [1, 0])
def test_fgsm_kernels_vs_oracle_on_a_full_size_batch(R, gdt, project):
    """64 x 3 x 224 x 224 (odd tail included), gradient as fp32 / bf16 / int8 signs, exact zeros in the gradient."""
    lib = R._lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 64 * 3 * 224 * 224 + 3
    x = torch.rand(n, device="cuda", generator=g) * 1.2 - 0.1
    t = torch.rand(n, device="cuda", generator=g)
    gr = torch.randn(n, device="cuda", generator=g)
    gr[::7] = 0
    eps, alpha, noise = 4 / 255, 1.25, 2.0
    start = torch.empty_like(x)
    S = torch.cuda.current_stream().cuda_stream
    assert lib.apgd_fgsm_start_f32(x.data_ptr(), t.data_ptr(), start.data_ptr(), n, eps, noise, project, S) == 0
    want_start = FO.fgsm_start(x.cpu().numpy(), t.cpu().numpy(), eps, True, noise, not project)
    assert np.array_equal(start.cpu().numpy(), want_start)
    gd = torch.sign(gr).to(torch.int8) if gdt == torch.int8 else gr.to(gdt)
    out = torch.empty_like(x)
    assert lib.apgd_fgsm_step_f32(x.data_ptr(), start.data_ptr(), gd.data_ptr(), R._lib.dtype_code(gdt), out.data_ptr(), n,
                                  float(alpha * eps), eps, project, S) == 0
    want = FO.fgsm_step(x.cpu().numpy(), want_start, gd.float().cpu().numpy(), eps, alpha, not project)
    assert np.array_equal(out.cpu().numpy(), want)
    # argument handling
    assert lib.apgd_fgsm_step_f32(x.data_ptr(), start.data_ptr(), gd.data_ptr(), 2, out.data_ptr(), n, 0.1, eps, 1, S) < 0      # fp16: no
    assert lib.apgd_fgsm_start_f32(x.data_ptr(), None, start.data_ptr(), n, eps, noise, 1, S) < 0
    assert lib.apgd_fgsm_start_f32(None, None, None, 0, eps, noise, 1, S) == 0


def test_fgsm_through_the_product_model_and_wrapped_model(R):
    """The selector's second branch end to end: WrappedModel(model, build_perturb(adv.attack=fgsm)) on the product ConvNeXt under
    bf16 autocast.  The attack receives int8 gradient signs from the stem kernel; replayed through the oracle from the recorded
    logits / signs and the same uniform draw it must match bit for bit."""
    from oracle import replay_tap  # noqa: F401  (same recording idea, one model call)
    torch.manual_seed(0)
    A = R.architecture
    cn = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), num_classes=10)
    cn.stem = A.ConvBlock1(48)
    cn = cn.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.rand(4, 3, 64, 64, device="cuda")
    y = torch.randint(0, 10, (4,), device="cuda")
    eps = 4 / 255
    rec = {}
    cls = R.ops.grad_sign_sink
    orig_exit = cls.__exit__

    def exit_and_record(self, *exc):
        rec["signs"] = None if self.signs is None else self.signs.clone()
        rec["fed"] = self.x_in.detach().clone()
        return orig_exit(self, *exc)

    torch.manual_seed(5)
    t = torch.rand_like(x)
    torch.manual_seed(5)
    try:
        cls.__exit__ = exit_and_record
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = R.fgsm_train(cn, x, y, eps, use_rs=True, alpha=1.25, noise_level=1.0)
    finally:
        cls.__exit__ = orig_exit
    assert rec["signs"] is not None and rec["signs"].dtype == torch.int8, "the int8 gradient-sign path was not taken"
    want, fed = FO.fgsm_train_oracle(O.ReplayModel(np.zeros((1, 4, 10), np.float32), rec["signs"].float().cpu().numpy()[None]),
                                     x.cpu().numpy(), y.cpu().numpy(), eps, t=t.cpu().numpy(), alpha=1.25, use_rs=True)
    assert np.array_equal(rec["fed"].cpu().numpy(), fed) and np.array_equal(out.cpu().numpy(), want)
    assert float((out - x).abs().max()) <= eps + 1e-7 and float(out.min()) >= 0 and float(out.max()) <= 1   # (x + d) - x: one ulp of x
    # through the boundary object, as the trainer builds it
    wm = R.wrap_model_for_at(cn, R.AdvConfig(attack="fgsm", eps=eps, alpha=1.25))
    assert isinstance(wm, R.WrappedModel)
    wm.set_perturb(True)
    wm.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = wm(x, y)
    assert logits.shape == (4, 10) and cn.training                     # the wrapper switched the model back to train mode
