#!/usr/bin/env python3
"""Generate model-surface fixtures from the REFERENCE classes (build container only).

Imports /root/reference/{utils_architecture.py, models/convnext.py, models/convnext_iso.py}
with a minimal stub for the absent ``timm`` package (SURVEY.md §8c recipe), instantiates small
versions of the reference's own modules, and stores weights + input + forward output +
input-gradient in tests/golden/model_*.npz.  Weight names are stored in the layout of
oracle/models_ref.py (timm key names; FB->timm map of SURVEY.md Appendix B)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_timm():
    class DropPath(nn.Identity):
        def __init__(self, p=0.):
            super().__init__()
    mods = {
        "timm": {}, "timm.models": {"create_model": None},
        "timm.models.layers": {"trunc_normal_": nn.init.trunc_normal_, "DropPath": DropPath},
        "timm.models.registry": {"register_model": lambda f: f},
        "timm.models.convnext": {"_create_convnext": None},
        "timm.models.vision_transformer": {"VisionTransformer": None},
    }
    for name, attrs in mods.items():
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    sys.modules["timm"].models = sys.modules["timm.models"]
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]          # bypass the broken models/__init__.py
    sys.modules["models"] = pkg


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_stub_timm()
cnx = _load("models.convnext", os.path.join(REF, "models/convnext.py"))
iso = _load("models.convnext_iso", os.path.join(REF, "models/convnext_iso.py"))
ua = _load("utils_architecture", os.path.join(REF, "utils_architecture.py"))


def fb_to_timm(sd):
    """FB ConvNeXt keys -> timm keys."""
    out = {}
    for k, v in sd.items():
        k2 = k
        if k.startswith("downsample_layers.0."):
            k2 = "stem." + k[len("downsample_layers.0."):]
        elif k.startswith("downsample_layers."):
            i = k.split(".")[1]
            k2 = f"stages.{i}.downsample." + k.split(".", 2)[2]
        elif k.startswith("stages."):
            _, i, j, rest = k.split(".", 3)
            rest = (rest.replace("dwconv", "conv_dw").replace("pwconv1", "mlp.fc1").replace("pwconv2", "mlp.fc2"))
            k2 = f"stages.{i}.blocks.{j}.{rest}"
        elif k.startswith("norm."):
            k2 = "head.norm." + k[5:]
        elif k.startswith("head."):
            k2 = "head.fc." + k[5:]
        out[k2] = v
    return out


def randomize(m, g):
    """Non-trivial LN weights/biases and gammas so every term is exercised."""
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1.0, p[0].numel()) ** 0.5))


def record(name, m, x, sd=None):
    m.eval()
    x = x.clone().requires_grad_()
    out = m(x)
    w = torch.linspace(-1, 1, out.numel()).reshape(out.shape)
    (gx,) = torch.autograd.grad((out * w).sum(), x)
    sd = sd if sd is not None else m.state_dict()
    blob = {f"w::{k}": v.detach().numpy() for k, v in sd.items()}
    blob.update(x=x.detach().numpy(), out=out.detach().numpy(), gx=gx.numpy(), cot=w.numpy())
    path = os.path.join(HERE, f"model_{name}.npz")
    np.savez_compressed(path, **blob)
    print(f"{name:18s} x{tuple(x.shape)} -> {tuple(out.shape)}  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(0)
    # channels-first LayerNorm (utils_architecture.py:57-81)
    ln = ua.LayerNorm(12, data_format="channels_first"); randomize(ln, g)
    record("ln_cf", ln, torch.randn(2, 12, 5, 7, generator=g) * 2 + 0.5)
    # ConvNeXt block with layer scale (models/convnext.py:15-50); weights stored under timm names
    blk = cnx.Block(16, layer_scale_init_value=1e-6); randomize(blk, g)
    sd = {k.replace("dwconv", "conv_dw").replace("pwconv1", "mlp.fc1").replace("pwconv2", "mlp.fc2"): v
          for k, v in blk.state_dict().items()}
    record("cn_block", blk, torch.randn(2, 16, 9, 9, generator=g), sd)
    blk0 = cnx.Block(24, layer_scale_init_value=0); randomize(blk0, g)          # iso variant: no gamma
    record("cn_block_nogamma", blk0, torch.randn(2, 24, 6, 6, generator=g))
    # conv stems
    for nm, mod, hw in (("stem_block1", ua.ConvBlock1(8), 16), ("stem_block3", ua.ConvBlock3(8), 16),
                        ("stem_block", ua.ConvBlock(4, end_siz=8, fin_dim=None), 32),
                        ("stem_block2", ua.ConvBlock2(4), 32)):
        randomize(mod, g)
        record(nm, mod, torch.rand(2, 3, hw, hw, generator=g))
    # ConvNeXt-iso-CvSt (models/convnext_iso.py:19-66 + ConvBlock stem, utils_architecture.py:237-239)
    m = iso.ConvNeXtIsotropic(depth=2, dim=32, num_classes=10)
    m.stem = ua.ConvBlock(4, end_siz=8, fin_dim=None); randomize(m, g)
    record("convnext_iso_cvst", m, torch.rand(2, 3, 32, 32, generator=g))
    # staged ConvNeXt with the ConvBlock1 stem (ConvNeXt-T-CvSt shape, utils_architecture.py:241-244)
    m = cnx.ConvNeXt(depths=[1, 1, 2, 1], dims=[8, 16, 32, 64], num_classes=10)
    m.downsample_layers[0] = ua.ConvBlock1(4); randomize(m, g)
    record("convnext_t_cvst", m, torch.rand(2, 3, 32, 32, generator=g), fb_to_timm(m.state_dict()))
    # ImageNormalizer wrapper (utils_architecture.py:86-117)
    nm = ua.normalize_model(ua.ConvBlock1(4), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)); randomize(nm.model, g)
    record("normalize_model", nm, torch.rand(1, 3, 8, 8, generator=g))


def pos_embed_fixture():
    """``interpolate_pos_encoding`` (utils_architecture.py:22-53): a 14x14(+cls) table resized for 320, 256 and 224."""
    g = torch.Generator().manual_seed(7)
    pe = torch.randn(1, 197, 12, generator=g)
    blob = dict(pos_embed=pe.numpy())
    for res in (320, 256, 224):
        blob[f"out_{res}"] = ua.interpolate_pos_encoding(pe, res, old_img_size=224, patch_size=16).numpy()
    path = os.path.join(HERE, "model_pos_embed.npz")
    np.savez_compressed(path, **blob)
    print("pos_embed", {k: v.shape for k, v in blob.items()}, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    if sys.argv[1:] == ["pos_embed"]:
        pos_embed_fixture()
    else:
        main()
        pos_embed_fixture()
