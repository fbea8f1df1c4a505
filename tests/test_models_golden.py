"""Pin oracle/models_ref.py (plain-torch restatement of the model surface) to fixtures produced
from the reference's own classes (tests/golden/make_model_golden.py; reference files
utils_architecture.py:57-217, models/convnext.py:15-117, models/convnext_iso.py:19-66)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import models_ref as M


def load(name):
    d = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w::")}
    return sd, torch.from_numpy(d["x"]), d["out"], d["gx"], torch.from_numpy(d["cot"])


def small_convnext_t():
    m = M.ConvNeXtTimm(depths=(1, 1, 2, 1), dims=(8, 16, 32, 64), num_classes=10)
    m.stem = M.ConvStem('block1', 4)
    return m


def small_iso():
    m = M.ConvNeXtIso(depth=2, dim=32, num_classes=10)
    m.stem = M.conv_block(4, end_siz=8, fin_dim=None)
    return m


BUILDERS = {
    "ln_cf": lambda: M.LayerNormCF(12),
    "cn_block": lambda: M.CNBlock(16),
    "cn_block_nogamma": lambda: M.CNBlock(24, ls_init=0, fb_names=True),
    "stem_block1": lambda: M.ConvStem('block1', 8),
    "stem_block3": lambda: M.ConvStem('block3', 8),
    "stem_block": lambda: M.conv_block(4, end_siz=8, fin_dim=None),
    "stem_block2": lambda: M.ConvStem('block2', 4),
    "convnext_iso_cvst": small_iso,
    "convnext_t_cvst": small_convnext_t,
    "normalize_model": lambda: M.normalize_model(M.ConvStem('block1', 4), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
}


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_models_ref_matches_reference(name):
    sd, x, out, gx, cot = load(name)
    m = BUILDERS[name]().eval()
    missing, unexpected = m.load_state_dict(sd, strict=True) if False else (None, None)
    m.load_state_dict(sd, strict=True)          # identical key names are part of the contract
    x = x.clone().requires_grad_()
    y = m(x)
    (g,) = torch.autograd.grad((y * cot).sum(), x)
    np.testing.assert_allclose(y.detach().numpy(), out, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(g.numpy(), gx, rtol=2e-4, atol=2e-6)


def test_param_counts_match_survey():
    # SURVEY.md §8 a13 [probe]: T-CvSt 28.63 M, iso-CvSt 23.04 M, B-CvSt 88.75 M, L-CvSt 198.13 M, ViT-B-CvSt ~87.15 M
    def n(m):
        return sum(p.numel() for p in m.parameters()) / 1e6
    assert abs(n(M.build('convnext_tiny')) - 28.63) < 0.01
    assert abs(n(M.build('convnext_iso')) - 23.04) < 0.01
    assert abs(n(M.build('convnext_base')) - 88.75) < 0.01
    assert abs(n(M.build('convnext_large')) - 198.13) < 0.01
    assert abs(n(M.build('vit_b')) - 87.15) < 0.05


def test_timm_key_layout():
    keys = set(M.build('convnext_tiny').state_dict())
    for k in ("stem.stem.0.weight", "stem.stem.1.bias", "stem.stem.4.weight", "stages.0.blocks.0.conv_dw.weight",
              "stages.1.downsample.0.weight", "stages.1.downsample.1.bias", "stages.2.blocks.8.mlp.fc2.bias",
              "stages.3.blocks.2.gamma", "stages.0.blocks.0.norm.weight", "head.norm.weight", "head.fc.bias"):
        assert k in keys, k
    assert not any(k.startswith("stages.0.downsample") for k in keys)
    keys = set(M.build('vit_b').state_dict())
    for k in ("cls_token", "pos_embed", "patch_embed.proj.stem.0.weight", "patch_embed.proj.stem.12.bias",
              "blocks.0.norm1.weight", "blocks.11.attn.qkv.bias", "blocks.3.attn.proj.weight", "blocks.5.mlp.fc1.weight",
              "norm.bias", "head.weight"):
        assert k in keys, k
    assert M.build('vit_b').pos_embed.shape == (1, 197, 768)
