"""CPU checks of the PRODUCT's model surface and host logic next to the hot path:

* state-dict keys and shapes of ``get_new_model(arch)`` equal the pinned oracle's for every arch name
  (``utils_architecture.py:225-322``) - "published checkpoints load";
* the product's module tree, run as the plain library composition (``ops.MODE = "eager"``, the only mode that accepts CPU
  tensors), reproduces the fixtures recorded from the REFERENCE's own classes (``tests/golden/make_model_golden.py``);
* ``interpolate_pos_encoding`` against a fixture produced by the reference function (``utils_architecture.py:22-53``);
* the LR table (``main.py:227-243, 956-958``), the EMA state dict / checkpoint files (``main.py:737-756, 882-887``);
* ``bench.py --gpus N`` never degrades silently to a single-rank run.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import revisiting_at_amd as R
from conftest import GOLDEN, ROOT
from oracle import models_ref as M
from revisiting_at_amd import checkpoint as CK

ARCHS = ["convnext_iso", "convnext_tiny", "convnext_small", "convnext_base", "convnext_large", "vit_s", "deit_s", "vit_m",
         "vit_b"]


@pytest.mark.parametrize("arch", ARCHS)
@pytest.mark.parametrize("not_original", [True, False])
def test_product_state_dict_keys_and_shapes_equal_the_oracle(arch, not_original):
    with torch.device("meta"):
        prod = R.get_new_model(arch, pretrained=False, not_original=not_original)
        ref = M.build(arch, not_original=not_original)
    ps, rs = prod.state_dict(), ref.state_dict()
    assert set(ps) == set(rs), (sorted(set(ps) - set(rs))[:5], sorted(set(rs) - set(ps))[:5])
    for k in ps:
        assert ps[k].shape == rs[k].shape, k


def _load_fixture(name):
    d = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w::")}
    return sd, torch.from_numpy(d["x"]), d["out"], d["gx"], torch.from_numpy(d["cot"])


def product_builders():
    A = R.architecture

    def small_t():
        m = A.ConvNeXt(depths=(1, 1, 2, 1), dims=(8, 16, 32, 64), num_classes=10)
        m.stem = A.ConvBlock1(4)
        return m

    def small_iso():
        m = A.ConvNeXtIsotropic(depth=2, dim=32, num_classes=10)
        m.stem = A.ConvBlock(4, end_siz=8, fin_dim=None)
        return m

    return {
        "ln_cf": lambda: A.LayerNorm(12, data_format="channels_first"),
        "cn_block": lambda: A.ConvNeXtBlock(16),
        "stem_block1": lambda: A.ConvBlock1(8),
        "stem_block3": lambda: A.ConvBlock3(8),
        "stem_block": lambda: A.ConvBlock(4, end_siz=8, fin_dim=None),
        "stem_block2": lambda: A.ConvBlock2(4),
        "convnext_iso_cvst": small_iso,
        "convnext_t_cvst": small_t,
        "normalize_model": lambda: A.normalize_model(A.ConvBlock1(4), A.IMAGENET_MEAN, A.IMAGENET_STD),
    }


@pytest.mark.parametrize("name", sorted(product_builders()))
def test_product_modules_as_library_composition_match_reference_fixtures(name, monkeypatch):
    monkeypatch.setattr(R.ops, "MODE", "eager")
    sd, x, out, gx, cot = _load_fixture(name)
    m = product_builders()[name]().eval()
    m.load_state_dict(sd, strict=True)
    x = x.clone().requires_grad_()
    y = m(x)
    (g,) = torch.autograd.grad((y * cot).sum(), x)
    np.testing.assert_allclose(y.detach().numpy(), out, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(g.numpy(), gx, rtol=2e-4, atol=2e-6)


def test_interpolate_pos_encoding_matches_reference_fixture():
    d = np.load(os.path.join(GOLDEN, "model_pos_embed.npz"))
    pe = torch.from_numpy(d["pos_embed"])
    for res in (320, 256, 224):
        got = R.architecture.interpolate_pos_encoding(pe, res, old_img_size=224, patch_size=16)
        assert got.shape == d[f"out_{res}"].shape
        np.testing.assert_allclose(got.numpy(), d[f"out_{res}"], rtol=1e-6, atol=1e-7)
    assert R.architecture.interpolate_pos_encoding(pe, 224) is pe                   # unchanged table is returned as is (:39-41)


def test_lr_table_known_answers():
    """``get_cosine_lr`` / per-iteration table (``main.py:227-243, 956-958``): linear warm-up from 1e-4*lr to lr over
    ``lr_peak_epoch`` epochs, then a half cosine to 5e-6; iterations interpolate linearly between epoch values."""
    from revisiting_at_amd.train_step import get_cosine_lr, iteration_lrs
    kw = dict(lr=1e-3, epochs=300, lr_peak_epoch=20)
    assert get_cosine_lr(0, **kw) == pytest.approx(1e-7, rel=1e-12)
    assert get_cosine_lr(10, **kw) == pytest.approx(0.5 * (1e-7 + 1e-3), rel=1e-12)
    assert get_cosine_lr(20, **kw) == pytest.approx(1e-3, rel=1e-12)
    assert get_cosine_lr(160, **kw) == pytest.approx(5e-6 + 0.5 * (1e-3 - 5e-6), rel=1e-12)     # cosine mid-point
    assert get_cosine_lr(300, **kw) == pytest.approx(5e-6, rel=1e-9)
    tab = iteration_lrs(5, 4, **kw)
    a, b = get_cosine_lr(5, **kw), get_cosine_lr(6, **kw)
    np.testing.assert_allclose(tab, [a, a + (b - a) / 4, a + (b - a) / 2, a + 3 * (b - a) / 4], rtol=1e-12)


def test_device_ema_state_dict_round_trips_through_the_checkpoint_files(tmp_path):
    """``weights_ema_{epoch}.pt`` / ``state_dict_ema`` (``main.py:739-747``) carry the wrapped model's keys and the
    ModelEmaV2 recursion ``ema = d*ema + (1-d)*model`` (``main.py:882-887, 996-997``)."""
    torch.manual_seed(0)
    base = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2))
    wrapped = R.WrappedModel(base, lambda m, x, y: x)
    ema = R.train_step.DeviceEma(wrapped, decay=0.9)
    p0 = {k: v.clone() for k, v in wrapped.state_dict().items()}
    with torch.no_grad():
        for p in wrapped.parameters():
            p.add_(1.0)
        base[1].num_batches_tracked.add_(5)
    ema.update()
    sd = ema.state_dict()
    assert list(sd) == list(wrapped.state_dict()) and all(k.startswith("base_model.") for k in sd)
    for k, v in wrapped.state_dict().items():
        if v.is_floating_point():
            expect = 0.9 * p0[k] + 0.1 * v
            torch.testing.assert_close(sd[k], expect, rtol=1e-6, atol=1e-7)
        else:
            assert torch.equal(sd[k], v)                                            # integer buffers are copied
    opt = torch.optim.AdamW(wrapped.parameters())
    CK.save_weights(wrapped, tmp_path, 1, ema_state=sd)
    CK.save_full(wrapped, opt, tmp_path, 5, ema_state=sd)
    fresh = R.WrappedModel(torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2)),
                           lambda m, x, y: x)
    assert CK.load_weights(fresh, str(tmp_path / "weights_ema_1.pt")) == 'standard loading'
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, sd[k])
    full = torch.load(tmp_path / "full_model_5.pth")
    assert set(full["state_dict_ema"]) == set(sd)
    ema2 = R.train_step.DeviceEma(fresh, decay=0.9)
    ema2.load_state_dict(full["state_dict_ema"])                                   # resume
    with pytest.raises(KeyError):
        ema2.load_state_dict({"nope": torch.zeros(1)})


def test_bench_refuses_to_run_fewer_ranks_than_requested():
    """``python bench.py --gpus 2`` must start 2 ranks or fail - never print a 1-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "--gpus 2" in r.stderr and "{" not in r.stdout
    # a torchrun-style environment whose world size disagrees with --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
