"""Plain-torch restatement of the reference's model surface.  TEST INFRASTRUCTURE ONLY.

Used (a) as the floating-point reference the fused HIP model kernels are compared with,
(b) as the model of the ``cpu_baseline`` leg of ``bench.py``.  Never imported by the product
package.  Eager torch ops only, no custom kernels, runs on any device.

What it restates and how it is pinned
-------------------------------------
* ``LayerNormCF`` / ``ConvStem*``  — ``/root/reference/utils_architecture.py:57-81, 120-217``
* ``CNBlock``                     — ``/root/reference/models/convnext.py:15-50`` (math of timm's
  ``ConvNeXtBlock``: dw7x7 -> LN -> Linear C->4C -> GELU -> Linear 4C->C -> gamma -> residual)
* ``ConvNeXtIso``                 — ``/root/reference/models/convnext_iso.py:19-66``
* ``ConvNeXtTimm``                — timm 0.8.0.dev0 ``ConvNeXt`` key layout
  built from the pieces above; numerically the FB ``ConvNeXt`` of ``models/convnext.py:52-117``
* ``ViTTimm``                     — timm 0.8.0.dev0 ``VisionTransformer``.
  timm is NOT in /root/reference: **parity unpinned** for the ViT body (only the ConvStem
  that replaces ``patch_embed.proj`` is pinned).

``tests/test_models_golden.py`` checks the pinned parts against fixtures produced by
``tests/golden/make_model_golden.py`` from the reference's own classes (forward output and
input-gradient, same weights).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNormCF(nn.Module):
    """channels_first LayerNorm, op for op as ``utils_architecture.py:76-81``."""

    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


def _stem_seq(chans, strides, final_1x1=None):
    """[conv3x3(s), LN_cf, GELU] * n (+ conv1x1) with the reference's Sequential indices."""
    layers = []
    for cin, cout, s in zip(chans[:-1], chans[1:], strides):
        layers += [nn.Conv2d(cin, cout, 3, stride=s, padding=1), LayerNormCF(cout), nn.GELU()]
    if final_1x1 is not None:
        layers.append(nn.Conv2d(chans[-1], final_1x1, 1))
    return nn.Sequential(*layers)


class ConvStem(nn.Module):
    """``ConvBlock`` (:120-144), ``ConvBlock1`` (:198-217), ``ConvBlock2`` (:146-171), ``ConvBlock3`` (:174-195).

    kind 'block'  : 3->p->2p->4p->8p (all s2) + 1x1 -> fin      (ViT-S/B, ConvNeXt-iso)
    kind 'block1' : 3->p->2p (s2, s2)                            (ConvNeXt-T/S)
    kind 'block2' : as 'block' with 1x1 -> 512                   (ViT-M)
    kind 'block3' : 3->p->1.5p->2p (s2, s2, s1)                  (ConvNeXt-B/L)
    Parameters live under ``stem.<idx>`` exactly as in the reference."""

    def __init__(self, kind, planes, fin_dim=None):
        super().__init__()
        p = planes
        if kind == 'block':
            self.stem = _stem_seq([3, p, 2 * p, 4 * p, 8 * p], [2, 2, 2, 2], final_1x1=fin_dim)
        elif kind == 'block1':
            self.stem = _stem_seq([3, p, 2 * p], [2, 2])
        elif kind == 'block2':
            self.stem = _stem_seq([3, p, 2 * p, 4 * p, 8 * p], [2, 2, 2, 2], final_1x1=512)
        elif kind == 'block3':
            self.stem = _stem_seq([3, p, int(p * 1.5), 2 * p], [2, 2, 1])
        else:
            raise ValueError(kind)

    def forward(self, x):
        return self.stem(x)


def conv_block(siz=48, end_siz=8, fin_dim=384):
    """``ConvBlock(siz, end_siz, fin_dim)``: fin = siz*end_siz unless fin_dim == 432 (:125)."""
    return ConvStem('block', siz, siz * end_siz if fin_dim != 432 else 432)


class CNBlock(nn.Module):
    """ConvNeXt block with timm parameter names (conv_dw, norm, mlp.fc1, mlp.fc2, gamma)."""

    def __init__(self, dim, ls_init=1e-6, fb_names=False):
        super().__init__()
        self.fb = fb_names
        dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        norm = nn.LayerNorm(dim, eps=1e-6)
        fc1, fc2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)
        if fb_names:                                   # models/convnext.py:28-32
            self.dwconv, self.norm, self.pwconv1, self.pwconv2 = dw, norm, fc1, fc2
        else:                                          # timm
            self.conv_dw, self.norm = dw, norm
            self.mlp = nn.Module()
            self.mlp.fc1, self.mlp.fc2 = fc1, fc2
        self.gamma = nn.Parameter(ls_init * torch.ones(dim)) if ls_init > 0 else None

    def parts(self):
        if self.fb:
            return self.dwconv, self.norm, self.pwconv1, self.pwconv2
        return self.conv_dw, self.norm, self.mlp.fc1, self.mlp.fc2

    def forward(self, x):
        dw, norm, fc1, fc2 = self.parts()
        y = dw(x).permute(0, 2, 3, 1)                  # :39-40
        y = fc2(F.gelu(fc1(norm(y))))                  # :41-44
        if self.gamma is not None:
            y = self.gamma * y                         # :45-46
        return x + y.permute(0, 3, 1, 2)               # :47-49


class LayerNorm2d(nn.LayerNorm):
    """timm ``LayerNorm2d``: LayerNorm over C of an NCHW tensor."""

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias,
                            self.eps).permute(0, 3, 1, 2)


class CNStage(nn.Module):
    def __init__(self, cin, cout, depth, first):
        super().__init__()
        self.downsample = nn.Identity() if first else nn.Sequential(LayerNorm2d(cin, eps=1e-6),
                                                                    nn.Conv2d(cin, cout, 2, stride=2))
        self.blocks = nn.Sequential(*[CNBlock(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ConvNeXtTimm(nn.Module):
    """timm-0.8 ``ConvNeXt`` key layout: stem / stages.i.{downsample,blocks.j} / head.{norm,fc}."""

    def __init__(self, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), num_classes=1000):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, dims[0], 4, stride=4), LayerNorm2d(dims[0], eps=1e-6))
        self.stages = nn.Sequential(*[CNStage(dims[max(i - 1, 0)], dims[i], depths[i], i == 0) for i in range(4)])
        self.head = nn.Module()
        self.head.norm = LayerNorm2d(dims[-1], eps=1e-6)
        self.head.fc = nn.Linear(dims[-1], num_classes)
        self.num_features = dims[-1]
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=.02)      # models/convnext.py:103-106
                nn.init.zeros_(m.bias)

    def forward_features(self, x):
        return self.stages(self.stem(x))

    def forward(self, x):
        x = self.forward_features(x).mean((-2, -1), keepdim=True)
        return self.head.fc(self.head.norm(x).flatten(1))


class ConvNeXtIso(nn.Module):
    """``ConvNeXtIsotropic`` (models/convnext_iso.py:19-66), FB parameter names."""

    def __init__(self, depth=18, dim=384, num_classes=1000):
        super().__init__()
        self.stem = nn.Conv2d(3, dim, 16, stride=16)
        self.blocks = nn.Sequential(*[CNBlock(dim, ls_init=0, fb_names=True) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.blocks(self.stem(x))
        return self.head(self.norm(x.mean((-2, -1))))


# ----------------------------------------------------------------------------- ViT (timm 0.8 layout; unpinned)
class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = ((q @ k.transpose(-2, -1)) * self.scale).softmax(-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _LS(nn.Module):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _VBlock(nn.Module):
    def __init__(self, dim, heads, init_values=None):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn, self.mlp = _Attn(dim, heads), _Mlp(dim)
        self.ls1 = _LS(dim, init_values) if init_values else nn.Identity()
        self.ls2 = _LS(dim, init_values) if init_values else nn.Identity()

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch, dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch, patch)
        self.grid_size = (img_size // patch, img_size // patch)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class ViTTimm(nn.Module):
    def __init__(self, dim=768, depth=12, heads=12, img_size=224, patch=16, num_classes=1000, init_values=None,
                 no_embed_class=False):
        super().__init__()
        self.patch_embed = _PatchEmbed(img_size, patch, dim)
        self.no_embed_class = no_embed_class
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        n = self.patch_embed.num_patches + (0 if no_embed_class else 1)
        self.pos_embed = nn.Parameter(torch.randn(1, n, dim) * .02)
        self.blocks = nn.Sequential(*[_VBlock(dim, heads, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        if self.no_embed_class:
            x = torch.cat([cls, x + self.pos_embed], 1)
        else:
            x = torch.cat([cls, x], 1) + self.pos_embed
        x = self.norm(self.blocks(x))
        return self.head(x[:, 0])


# ----------------------------------------------------------------------------- arch names (utils_architecture.py:225-322)
def build(modelname: str, not_original: bool = True, img_size: int = 224):
    if modelname == 'convnext_iso':                                   # commented recipe :235-239
        m = ConvNeXtIso()
        if not_original:
            m.stem = conv_block(48, end_siz=8, fin_dim=384)
    elif modelname in ('convnext_tiny', 'convnext_small'):            # :241-254
        m = ConvNeXtTimm((3, 3, 9, 3) if modelname == 'convnext_tiny' else (3, 3, 27, 3))
        if not_original:
            m.stem = ConvStem('block1', 48)
    elif modelname == 'convnext_base':                                # :256-262
        m = ConvNeXtTimm((3, 3, 27, 3), (128, 256, 512, 1024))
        if not_original:
            m.stem = ConvStem('block3', 64)
    elif modelname == 'convnext_large':                               # :264-269
        m = ConvNeXtTimm((3, 3, 27, 3), (192, 384, 768, 1536))
        if not_original:
            m.stem = ConvStem('block3', 96)
    elif modelname in ('vit_s', 'deit_s'):                            # :271-284
        m = ViTTimm(384, 12, 6, img_size)
        if not_original:
            m.patch_embed.proj = conv_block(48, end_siz=8)
    elif modelname == 'vit_m':                                        # :286-291
        m = ViTTimm(512, 12, 8, img_size, init_values=1e-6, no_embed_class=True)
        if not_original:
            m.patch_embed.proj = ConvStem('block2', 48)
    elif modelname == 'vit_b':                                        # :297-301
        m = ViTTimm(768, 12, 12, img_size)
        if not_original:
            m.patch_embed.proj = conv_block(48, end_siz=16, fin_dim=None)
    else:
        raise ValueError(modelname)
    return m


class ImageNormalizer(nn.Module):
    """``(x - mean) / std`` — utils_architecture.py:86-98."""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer('mean', torch.as_tensor(mean).view(1, 3, 1, 1))
        self.register_buffer('std', torch.as_tensor(std).view(1, 3, 1, 1))

    def forward(self, x):
        return (x - self.mean) / self.std


def normalize_model(model, mean, std):
    """``nn.Sequential(normalize=..., model=...)`` — utils_architecture.py:111-117."""
    return nn.Sequential(OrderedDict([('normalize', ImageNormalizer(mean, std)), ('model', model)]))
