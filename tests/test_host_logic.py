"""Host-side logic of the product package that needs no GPU: schedule, flag parsing, the
WrappedModel boundary (main.py:260-301) and the no-CPU-fallback rule."""
import functools

import pytest
import torch
import torch.nn as nn

import revisiting_at_amd as R
from oracle import apgd_oracle as O


@pytest.mark.parametrize("n_iter", list(range(1, 60)) + [100, 250, 1000])
def test_schedule_matches_oracle(n_iter):
    assert R.checkpoint_schedule(n_iter) == O.checkpoint_schedule(n_iter)


def test_schedule_k_never_exceeds_history():
    # the device state machine reads loss_steps[i-k+1 .. i]; the C ABI rejects k > i+1
    for n in range(1, 300):
        for i, k in R.checkpoint_schedule(n):
            assert 1 <= k <= i + 1 < n + 1


def test_adv_flags_defaults_and_parsing():
    c = R.AdvConfig()
    assert (c.attack, c.norm, c.n_iter) == ("none", "Linf", 2) and abs(c.eps - 4 / 255) < 1e-12   # main.py:175-184
    c = R.AdvConfig.from_argv("--data.num_workers 4 --adv.attack apgd --adv.n_iter=3 --adv.norm L2 --adv.eps 0.5 "
                              "--adv.verbose 1 --lr.lr 1e-3".split())
    assert (c.attack, c.n_iter, c.norm, c.eps, c.verbose) == ("apgd", 3, "L2", 0.5, 1)
    assert abs(R.AdvConfig.from_argv(["--adv.eps", "8/255"]).eps - 8 / 255) < 1e-12
    with pytest.raises(ValueError):
        R.AdvConfig.from_argv(["--adv.bogus", "1"])


def test_build_perturb_wiring():
    assert R.build_perturb(R.AdvConfig()) is None                       # attack 'none': model is not wrapped
    mix = object()
    p = R.build_perturb(R.AdvConfig(attack="apgd", n_iter=5, eps=0.1, norm="L2", verbose=1), mixup=mix)
    assert isinstance(p, functools.partial) and p.func is R.apgd_train  # main.py:834-835
    assert p.keywords == dict(norm="L2", eps=0.1, n_iter=5, verbose=True, mixup=mix)
    f = R.build_perturb(R.AdvConfig(attack="fgsm", eps=0.02, alpha=1.25, noise_level=2.0, skip_projection=1))
    assert isinstance(f, functools.partial) and f.func is R.fgsm_train  # main.py:836-842
    assert f.keywords == dict(eps=0.02, use_rs=True, alpha=1.25, noise_level=2.0, skip_projection=True)
    with pytest.raises(ValueError):
        R.build_perturb(R.AdvConfig(attack="pgd"))
    m = nn.Linear(3, 2)
    assert R.wrap_model_for_at(m, R.AdvConfig()) is m


def test_wrapped_model_protocol_on_cpu():
    calls = []

    class Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.l = nn.Linear(4, 3)

        def forward(self, x):
            calls.append(("fwd", self.training))
            return self.l(x)

    def perturb(model, x, y):
        calls.append(("perturb", model.training))
        return (x + 1.0, "acc", "loss", "x_best_adv")

    wm = R.WrappedModel(Base(), perturb)
    assert list(wm.state_dict()) == ["base_model.l.weight", "base_model.l.bias"]
    x = torch.zeros(2, 4)
    wm.train()
    assert torch.equal(wm(x), wm.base_model.l(x)) and calls == [("fwd", True)]
    wm.set_perturb(True)
    with pytest.raises(AssertionError):
        wm(x)
    calls.clear()
    out = wm(x, torch.zeros(2, dtype=torch.long))
    assert calls == [("perturb", False), ("fwd", True)]                 # eval during the attack, train after
    assert torch.equal(out, wm.base_model.l(x + 1.0))                   # z[0] is what gets trained on
    wm.perturb = lambda m, a, b: a + 2.0                                # a bare tensor is accepted too (main.py:291)
    assert torch.equal(wm(x, 0), wm.base_model.l(x + 2.0))


def test_apgd_train_has_no_cpu_fallback():
    m = nn.Linear(4, 3).eval()
    x, y = torch.rand(2, 4), torch.zeros(2, dtype=torch.long)
    with pytest.raises(R._lib.ApgdHipError, match="no CPU fallback"):
        R.apgd_train(m, x, y, norm="Linf", eps=0.1, n_iter=2)
    with pytest.raises(AssertionError):
        R.apgd_train(m.train(), x, y, norm="Linf", eps=0.1)
    with pytest.raises(KeyError):
        R.apgd_train(m.eval(), x, y, norm="Linf", eps=0.1, loss="hinge")


def test_product_package_never_imports_the_oracle():
    import os
    import re
    pkg = os.path.dirname(R.__file__)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_fgsm_train_has_no_cpu_fallback_and_keeps_the_reference_errors():
    m = nn.Linear(3, 2).eval()
    x, y = torch.rand(2, 3), torch.tensor([0, 1])
    with pytest.raises(R._lib.ApgdHipError):
        R.fgsm_train(m, x, y, 0.1)                                        # CPU tensor: no fallback in the product
    with pytest.raises(KeyError):
        R.fgsm_train(m, x, y, 0.1, loss="dlr")                            # criterion_dict of fgsm_train.py has 'ce' only
    with pytest.raises(AssertionError):
        R.fgsm_train(m.train(), x, y, 0.1)                                # fgsm_train.py:74


def test_vit_block_operators_fall_back_to_the_plain_composition_off_the_device():
    """``ops.layer_norm_skip`` / ``ops.mlp_residual`` / ``ops.linear_lib`` (the ViT block's fused forms) dispatch to the plain
    PyTorch composition for CPU tensors - the model surface stays importable and testable without a GPU; the ATTACK has no CPU
    path (``test_apgd_train_has_no_cpu_fallback``)."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    C = 16
    x = torch.randn(2, 5, C, requires_grad=True)
    w, b = torch.randn(C), torch.randn(C)
    y, xs = R.ops.layer_norm_skip(x, w, b, 1e-6)
    assert xs is x and torch.equal(y, F.layer_norm(x, (C,), w, b, 1e-6))
    w1, b1, w2, b2 = torch.randn(4 * C, C), torch.randn(4 * C), torch.randn(C, 4 * C), torch.randn(C)
    g = torch.randn(C)
    h = torch.randn(2, 5, C)
    ref = x + g * F.linear(F.gelu(F.linear(h, w1, b1)), w2, b2)
    assert torch.allclose(R.ops.mlp_residual(x, h, w1, b1, w2, b2, g), ref, atol=1e-6)
    assert torch.allclose(R.ops.mlp_residual(x, h, w1, b1, w2, b2, None), x + F.linear(F.gelu(F.linear(h, w1, b1)), w2, b2), atol=1e-6)
    assert torch.equal(R.ops.linear_lib(h, w1, b1), F.linear(h, w1, b1))
    blk = R.architecture.VitBlock(C, 2, init_values=0.1).eval()
    out = blk(x)
    (gx,) = torch.autograd.grad(out.sum(), x)
    assert out.shape == x.shape and torch.isfinite(gx).all()


def test_evaluation_batch_buckets():
    """aa_eval._bucket: a handful of padded sizes per evaluation batch size (bs, then halves, not below 8), never below the subset."""
    from revisiting_at_amd import aa_eval
    for bs in (100, 200, 32, 7):
        sizes = {aa_eval._bucket(k, bs) for k in range(1, bs + 1)}
        assert len(sizes) <= 6 and max(sizes) == bs and all(aa_eval._bucket(k, bs) >= k for k in range(1, bs + 1))
    assert aa_eval._bucket(150, 100) == 150                            # (a subset larger than bs cannot happen; it is not truncated)


def test_kernel_sets_switch_in_a_running_process_and_restore():
    """bench.py's A/B leg: the whole kernel selection (module switches + the two inside the library) flips at run time and comes back."""
    lib = R._lib.load()
    ops, apgd = R.ops, R.apgd
    start = ops.kernel_set("default")
    try:
        assert lib.cnx_runtime_switch(0, -1) == 3 and lib.cnx_runtime_switch(1, -1) == 1 and lib.cnx_runtime_switch(99, 1) == -1
        assert lib.cnx_runtime_switch(3, -1) == 3 and ops.kernel_set("round5")["blk2b"] == 3 and lib.cnx_runtime_switch(3, -1) == 0
        ops.kernel_set("default")
        prev = ops.kernel_set("round4")
        assert prev == ops.KERNEL_SETS["default"]
        assert (ops._WGRAD_MODE, ops.STEM_WGRAD_HIP, ops._TRAIN_HPRE_WIDTHS) == ("lib", False, set())
        assert (ops._DGAMMA_FROM_DW2, ops._DLN_FROM_DW1, ops._LN_IN_TRAIN_BWD, ops._POOL_ROWS) == (False, False, False, False)
        assert apgd.FUSED_TRACKING is False and ops._TN_PAIR is False
        assert lib.cnx_runtime_switch(0, -1) == 0 and lib.cnx_runtime_switch(1, -1) == 0 and lib.cnx_runtime_switch(3, -1) == 0
        assert ops.kernel_set(prev) == dict(ops.KERNEL_SETS["default"], **ops.KERNEL_SETS["round4"])   # (a set names only what it changes)
        assert lib.cnx_runtime_switch(2, 3) == -1 and ops.kernel_set("w8")["fwd_w8"] == 0   # measurement builds only: a no-op here
        assert ops.kernel_set({"dln": "kernel"})["dln"] == "dw1"
        assert (ops._DLN_FROM_DW1, ops._LN_IN_TRAIN_BWD) == (True, False)
        with pytest.raises(ValueError):
            ops.kernel_set({"no_such_switch": 1})
        with pytest.raises(ValueError):
            ops.kernel_set({"wgrad": "cuda"})
    finally:
        ops.kernel_set(start)
    assert ops.kernel_set({}) == start
