"""Host-side pieces next to the hot path: Mixup/CutMix soft targets and checkpoint key handling."""
import numpy as np
import pytest
import torch

import revisiting_at_amd as R
from revisiting_at_amd import checkpoint as CK
from revisiting_at_amd.mixup import Mixup, SoftTargetCrossEntropy, mixup_target


def test_mixup_targets_and_images():
    torch.manual_seed(0)
    x = torch.rand(8, 3, 16, 16)
    y = torch.randint(0, 10, (8,))
    mx = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=10, seed=3)
    seen = set()
    for _ in range(12):
        xm, ym = mx(x, y)
        assert xm.shape == x.shape and ym.shape == (8, 10) and ym.dtype == torch.float32
        assert torch.allclose(ym.sum(1), torch.ones(8), atol=1e-6) and float(ym.min()) >= 0
        assert float(xm.min()) >= 0 and float(xm.max()) <= 1                      # convex / pasted: stays in the box
        changed = (xm != x).flatten(1).any(1)
        is_cutmix = bool(((xm == x) | (xm == x.flip(0))).all())
        seen.add(is_cutmix)
        if is_cutmix:                                                              # area-corrected lam = kept fraction
            lam = float((xm == x).float().mean()) if changed.any() else 1.0
            on = 1 - 0.1 + 0.01
            assert abs(float(ym[0, y[0]]) - (on * lam + (on if y[0] == y[-1] else 0.01) * (1 - lam))) < 0.05
    assert seen == {True, False}                                                   # both branches were drawn
    with pytest.raises(ValueError):
        mx(x[:7], y[:7])
    t = mixup_target(torch.tensor([1, 2]), 4, lam=0.25, smoothing=0.0)
    assert torch.allclose(t, torch.tensor([[0, .25, .75, 0], [0, .75, .25, 0]]))


def test_soft_target_ce_matches_hard_ce_on_one_hot():
    torch.manual_seed(1)
    z = torch.randn(5, 7)
    y = torch.randint(0, 7, (5,))
    soft = SoftTargetCrossEntropy()(z, torch.nn.functional.one_hot(y, 7).float())
    assert torch.allclose(soft, torch.nn.functional.cross_entropy(z, y), atol=1e-6)


def test_checkpoint_fallback_chain_and_eval_cleanup(tmp_path):
    torch.manual_seed(0)
    base = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    wrapped = R.WrappedModel(torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2)), lambda m, x, y: x)
    ddp_style = {f"module.{k}": v for k, v in wrapped.state_dict().items()}          # what main.py:738 saves under DDP
    assert CK.load_weights(wrapped, ddp_style) == 'standard loading'
    assert CK.load_weights(base, ddp_style) == 'loaded'                                # strip base_model.
    assert CK.load_weights(wrapped, base.state_dict()) == 'loaded from clean model'    # add base_model.
    p = tmp_path / "w.pt"
    torch.save(ddp_style, p)
    assert CK.load_weights(base, str(p)) == 'loaded'
    for a, b in zip(base.state_dict().values(), wrapped.state_dict().values()):
        assert torch.equal(a, b)
    cleaned = CK.clean_eval_keys({"module.base_model.stem.0.weight": 1, "module.base_model.se_fc.weight": 2})
    assert set(cleaned) == {"stem.0.weight", "se_module.fc.weight"}
    opt = torch.optim.AdamW(base.parameters())
    CK.save_weights(base, tmp_path, 3, ema_state=base.state_dict())
    CK.save_full(base, opt, tmp_path, 5)
    assert (tmp_path / "weights_3.pt").exists() and (tmp_path / "weights_ema_3.pt").exists()
    full = torch.load(tmp_path / "full_model_5.pth")
    assert set(full) == {"model_state_dict", "optimizer_state_dict", "epoch"} and full["epoch"] == 5
    with pytest.raises(RuntimeError):
        CK.load_weights(base, {"nope.weight": torch.zeros(1)})
