"""timm-compatible model surface of the reference (``/root/reference/utils_architecture.py``).

``get_new_model(modelname, pretrained, not_original, updated)`` keeps the reference's
signature and arch names (``utils_architecture.py:225-322``) and returns modules whose
state-dict keys are timm 0.8's, so the published checkpoints load:

  ConvNeXt  : stem.* | stages.{i}.downsample.{0,1}.* | stages.{i}.blocks.{j}.{conv_dw,norm,mlp.fc1,mlp.fc2}.*,
              .gamma | head.norm.* | head.fc.*          ("CvSt": stem.stem.{0,1,3,4[,6,7]}.*)
  ViT       : cls_token, pos_embed | patch_embed.proj.* | blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*
              | norm.* | head.*                          ("CvSt": patch_embed.proj.stem.{0,...,12}.*)

timm itself is not a dependency.  The heavy per-block arithmetic is routed through
``revisiting_at_amd.ops`` (fused gfx950 kernels where they exist, PyTorch-ROCm library calls —
MIOpen / hipBLASLt — for plain convolutions and GEMMs otherwise).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

__all__ = ["get_new_model", "LayerNorm", "ConvBlock", "ConvBlock1", "ConvBlock2", "ConvBlock3", "ConvNeXt",
           "ConvNeXtBlock", "ConvNeXtIsotropic", "VisionTransformer", "ImageNormalizer", "normalize_model",
           "interpolate_pos_encoding", "IMAGENET_MEAN", "IMAGENET_STD"]

IMAGENET_MEAN = (0.485, 0.456, 0.406)       # main.py:191-192 (as fractions of 255 there)
IMAGENET_STD = (0.229, 0.224, 0.225)


# ----------------------------------------------------------------------------- norms
class LayerNorm(nn.Module):
    """LayerNorm over C for channels_last ``[N,H,W,C]`` or channels_first ``[N,C,H,W]`` inputs
    (``utils_architecture.py:57-81``; default data_format there is channels_first)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_first"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return ops.layer_norm_cf(x, self.weight, self.bias, self.eps)


class LayerNorm2d(nn.LayerNorm):
    """timm's LayerNorm2d (NCHW in, NCHW out, normalised over C)."""

    def forward(self, x):
        return ops.layer_norm_cf(x, self.weight, self.bias, self.eps)


class ImageNormalizer(nn.Module):
    """``(input - mean) / std`` (``utils_architecture.py:86-98``)."""

    def __init__(self, mean: Tuple[float, float, float], std: Tuple[float, float, float], persistent: bool = True):
        super().__init__()
        self.register_buffer('mean', torch.as_tensor(mean).view(1, 3, 1, 1), persistent=persistent)
        self.register_buffer('std', torch.as_tensor(std).view(1, 3, 1, 1), persistent=persistent)

    def forward(self, input):
        return (input - self.mean) / self.std


def normalize_model(model: nn.Module, mean, std) -> nn.Module:
    """Prefixes: ``normalize.{mean,std}`` and ``model.*`` (``utils_architecture.py:111-117``)."""
    return nn.Sequential(OrderedDict([('normalize', ImageNormalizer(mean, std)), ('model', model)]))


# ----------------------------------------------------------------------------- ConvStem ("CvSt")
class _CfLnGelu(nn.Module):
    """LN(channels_first) followed by GELU, kept as ONE module at the LayerNorm's Sequential
    index so the parameter keys stay ``stem.<idx>.{weight,bias}``; the following ``nn.Identity``
    holds the GELU's index.  Runs as a single fused kernel."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = 1e-6

    def forward(self, x):
        return ops.layer_norm_cf_gelu(x, self.weight, self.bias, self.eps)


class _ImageConv(nn.Conv2d):
    """The first stem convolution (3 -> P, 3x3, stride 2): a plain ``nn.Conv2d`` for the state dict, routed through the
    hand-written image-to-NHWC kernel (and its input-gradient kernel, which produces the attack's gradient) when the input
    is the fp32 NCHW image batch under bf16 autocast."""

    def forward(self, x):
        if ops.stem_conv_supported(x, self.weight, self.stride, self.padding):
            return ops.stem_conv(x, self.weight, self.bias)
        return ops.conv2d_lib(x, self)


class _StemConv2(nn.Conv2d):
    """A later ConvStem convolution (3x3, stride 2: ``ConvBlock1`` 48 -> 96, ``ConvBlock3`` 64 -> 96): plain ``nn.Conv2d`` for the
    state dict, routed through the hand-written implicit-GEMM kernel when its input is a channels-last bf16 activation."""

    def forward(self, x):
        if ops.conv3x3s2_supported(x, self):
            return ops.conv3x3s2(x, self.weight, self.bias)
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and self.out_channels % 16:
            # Library guard: MIOpen's fp32 NHWC backward-data implicit-GEMM solver reads past the end of its operands at
            # some narrow widths (8 -> 12 and 16 -> 24 on 8x8 maps fault when the tensor ends a mapped segment:
            # profiles/r02_miopen_nhwc_bwd_fault.md, tools/probe/conv_fault_fuzz.py); its NCHW solvers do not.
            x = x.contiguous()
        return ops.conv2d_lib(x, self)


class _LibConv(nn.Conv2d):
    """A ConvStem convolution that always runs in the library (stride 1, the 1x1 projection): a plain ``nn.Conv2d`` for the state dict
    whose backward takes the input and filter gradients from the library and sums the bias gradient itself (``ops.conv_bias_grad``:
    the library's came back non-finite under hipGraph replay of the training pass)."""

    def forward(self, x):
        return ops.conv2d_lib(x, self)


class _StemSequential(nn.Sequential):
    """``nn.Sequential`` (same indices / state-dict keys) that runs the first convolution and its LayerNorm + GELU as one
    kernel when the input is the fp32 image batch under bf16 autocast."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, _ImageConv) and i + 1 < len(mods) and isinstance(mods[i + 1], _CfLnGelu)
                    and ops.stem_conv_supported(x, m.weight, m.stride, m.padding) and ops.stem_fused_ln()):
                n = mods[i + 1]
                x = ops.stem_conv_ln_gelu(x, m.weight, m.bias, n.weight, n.bias, n.eps)
                i += 2
                continue
            x = m(x)
            i += 1
        return x


def _stem(chans, strides, final_1x1=None):
    layers = []
    for cin, cout, s in zip(chans[:-1], chans[1:], strides):
        conv = _ImageConv if (cin == 3 and s == 2) else (_StemConv2 if s == 2 else _LibConv)
        layers += [conv(cin, cout, kernel_size=3, stride=s, padding=1), _CfLnGelu(cout), nn.Identity()]
    if final_1x1 is not None:
        layers.append(_LibConv(chans[-1], final_1x1, kernel_size=1, stride=1, padding=0))
    return _StemSequential(*layers)


class ConvBlock(nn.Module):
    """4x (conv3x3 s2, LN_cf, GELU) + conv1x1 (``utils_architecture.py:120-144``): ViT / iso stem."""
    expansion = 1

    def __init__(self, siz=48, end_siz=8, fin_dim=384):
        super().__init__()
        self.planes = p = siz
        fin_dim = p * end_siz if fin_dim != 432 else 432
        self.stem = _stem([3, p, p * 2, p * 4, p * 8], [2, 2, 2, 2], final_1x1=fin_dim)

    def forward(self, x):
        return self.stem(x)


class ConvBlock2(nn.Module):
    """As ConvBlock with a fixed 512-wide projection (``:146-171``); DeiT-III medium only."""
    expansion = 1

    def __init__(self, siz=48, end_siz=8, fin_dim=384):
        super().__init__()
        self.planes = p = siz
        self.stem = _stem([3, p, p * 2, p * 4, p * 8], [2, 2, 2, 2], final_1x1=512)

    def forward(self, x):
        return self.stem(x)


class ConvBlock3(nn.Module):
    """conv s2, conv s2, conv s1 with LN_cf+GELU each (``:174-195``): ConvNeXt-B/L stem."""

    def __init__(self, siz=64):
        super().__init__()
        self.planes = p = siz
        self.stem = _stem([3, p, int(p * 1.5), p * 2], [2, 2, 1])

    def forward(self, x):
        return self.stem(x)


class ConvBlock1(nn.Module):
    """conv s2, conv s2 with LN_cf+GELU each (``:198-217``): ConvNeXt-T/S stem."""

    def __init__(self, siz=48, end_siz=8, fin_dim=384):
        super().__init__()
        self.planes = p = siz
        self.stem = _stem([3, p, p * 2], [2, 2])

    def forward(self, x):
        return self.stem(x)


# ----------------------------------------------------------------------------- ConvNeXt
class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtBlock(nn.Module):
    """``x + gamma * fc2(GELU(fc1(LN(dw7x7(x)))))`` (``models/convnext.py:15-50``; timm names)."""

    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, 4 * dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim)) if ls_init_value > 0 else None

    def forward(self, x):
        return ops.convnext_block(x, self.conv_dw.weight, self.conv_dw.bias, self.norm.weight, self.norm.bias,
                                  self.norm.eps, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                                  self.mlp.fc2.bias, self.gamma)


class ConvNeXtStage(nn.Module):
    def __init__(self, in_chs, out_chs, depth, first, ls_init_value=1e-6):
        super().__init__()
        if first:
            self.downsample = nn.Identity()
        else:
            self.downsample = nn.Sequential(LayerNorm2d(in_chs, eps=1e-6),
                                            _LibConv(in_chs, out_chs, kernel_size=2, stride=2))   # (library path: widths that are no multiple of 24)
        self.blocks = nn.Sequential(*[ConvNeXtBlock(out_chs, ls_init_value) for _ in range(depth)])

    def forward(self, x):
        ds = self.downsample
        if isinstance(ds, nn.Sequential) and ops.downsample_supported(x, ds[0].weight, ds[1]):
            # LayerNorm written in 2x2-patch form + one library GEMM with the bias in its epilogue
            x = ops.downsample_ln_conv(x, ds[0].weight, ds[0].bias, ds[0].eps, ds[1].weight, ds[1].bias)
        else:
            x = ds(x)
        return self.blocks(x)


class _Head(nn.Module):
    def __init__(self, dim, num_classes):
        super().__init__()
        self.norm = LayerNorm2d(dim, eps=1e-6)
        self.fc = nn.Linear(dim, num_classes)

    def forward(self, x):
        x = ops.global_pool(x)                       # global_pool
        h = self.norm(x).flatten(1)
        if (ops._ATTACK_PASS and h.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            # inside the attack of a two-stream model no GEMM may be the library's (graphed._streams): the classifier too runs
            # on cnx_gemm_nt (autocast would hand nn.Linear the same bf16 operands)
            return ops.linear_lib(h.to(torch.bfloat16).contiguous(), self.fc.weight, self.fc.bias)
        return self.fc(h)


class ConvNeXt(nn.Module):
    """timm-0.8 ConvNeXt: ``forward = head(stages(stem(x)))``; ``model.stem`` is what the CvSt
    variants replace wholesale (``utils_architecture.py:243-244, 260-262, 268-269``)."""

    def __init__(self, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), num_classes=1000, ls_init_value=1e-6):
        super().__init__()
        self.num_classes, self.num_features = num_classes, dims[-1]
        # graphed.py: a captured attack on this model runs as two batch chunks on two streams.  Pays on the narrow pyramids
        # (ConvNeXt-T / -S: the late stages' kernels have 98 - 392 workgroups for 256 CUs); the wide ones and the isotropic /
        # transformer models are GEMM-bound at every depth and keep one stream (and the library's GEMMs, measured faster there)
        self.apgd_two_streams = max(dims) <= 768
        self.stem = nn.Sequential(_LibConv(3, dims[0], kernel_size=4, stride=4), LayerNorm2d(dims[0], eps=1e-6))
        self.stages = nn.Sequential(*[ConvNeXtStage(dims[max(i - 1, 0)], dims[i], depths[i], i == 0, ls_init_value)
                                      for i in range(4)])
        self.norm_pre = nn.Identity()
        self.head = _Head(dims[-1], num_classes)
        self.apply(_init_convnext)

    def forward_features(self, x):
        return self.norm_pre(self.stages(self.stem(x)))

    def forward_head(self, x):
        return self.head(x)

    def ddp_cut(self):
        """``train_step.FlatGradSync``: (module whose output splits the backward, modules behind it).  Stages 2, 3 and the head hold
        95 % of ConvNeXt-T's parameter bytes and their gradients are complete after less than half of the backward: their
        all-reduce runs under the backward of stages 1, 0 and the stem."""
        return self.stages[1], [self.stages[2], self.stages[3], self.norm_pre, self.head]

    def forward(self, x):
        return self.forward_head(self.forward_features(x))


def _init_convnext(m):
    if isinstance(m, (nn.Conv2d, nn.Linear)):                 # models/convnext.py:103-106
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class _IsoBlock(nn.Module):
    """FB-named block (dwconv / norm / pwconv1 / pwconv2, no gamma) of the isotropic model."""

    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return ops.convnext_block(x, self.dwconv.weight, self.dwconv.bias, self.norm.weight, self.norm.bias,
                                  self.norm.eps, self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight,
                                  self.pwconv2.bias, None)


class ConvNeXtIsotropic(nn.Module):
    """``ConvNeXtIsotropic`` (``models/convnext_iso.py:19-66``): stem, 18 gamma-less blocks, pooled LN, head."""

    def __init__(self, in_chans=3, num_classes=1000, depth=18, dim=384):
        super().__init__()
        self.stem = _LibConv(in_chans, dim, kernel_size=16, stride=16)
        self.blocks = nn.Sequential(*[_IsoBlock(dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        self.apply(_init_convnext)

    def forward_features(self, x):
        return self.norm(self.blocks(self.stem(x)).mean((-2, -1)))

    def ddp_cut(self):
        k = len(self.blocks) // 2                     # equal blocks: the second half's gradients go out under the first half's backward
        return self.blocks[k - 1], [*self.blocks[k:], self.norm, self.head]

    def forward(self, x):
        return self.head(self.forward_features(x))


# ----------------------------------------------------------------------------- ViT (timm 0.8 VisionTransformer)
class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = ops.linear_lib(x, self.qkv.weight, self.qkv.bias)
        o = ops.attention(qkv, self.num_heads, self.scale)      # [B, N, C]
        return ops.linear_lib(o, self.proj.weight, self.proj.bias)


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _RowLayerNorm(nn.LayerNorm):
    """``nn.LayerNorm`` (same parameters / state-dict keys) on the hand-written row kernel."""

    def forward(self, x):
        return ops.layer_norm_last(x, self.weight, self.bias, self.eps)


class VitBlock(nn.Module):
    def __init__(self, dim, num_heads, init_values=None):
        super().__init__()
        self.norm1 = _RowLayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.ls1 = _LayerScale(dim, init_values) if init_values else nn.Identity()
        self.norm2 = _RowLayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, 4 * dim)
        self.ls2 = _LayerScale(dim, init_values) if init_values else nn.Identity()

    def forward(self, x):
        g1 = self.ls1.gamma if isinstance(self.ls1, _LayerScale) else None
        g2 = self.ls2.gamma if isinstance(self.ls2, _LayerScale) else None
        # x + ls1(attn(norm1(x))), x + ls2(mlp(norm2(x))): the LayerNorm hands x back for the skip connection so that its
        # backward kernel sums both gradients of x (ops.layer_norm_skip)
        h, xs = ops.layer_norm_skip(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = ops.scale_residual(xs, self.attn(h), g1)
        h, xs = ops.layer_norm_skip(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        m = self.mlp
        return ops.mlp_residual(xs, h, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, g2)


class PatchEmbed(nn.Module):
    """Attributes used by the eval harness: proj, patch_size, img_size, num_patches, grid_size
    (``AA_eval.py:198-211``)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = _LibConv(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class VisionTransformer(nn.Module):

    def __init__(self, img_size=224, patch_size=16, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 init_values=None, no_embed_class=False):
        super().__init__()
        self.num_classes, self.embed_dim, self.no_embed_class = num_classes, embed_dim, no_embed_class
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        n_pos = self.patch_embed.num_patches + (0 if no_embed_class else 1)
        self.pos_embed = nn.Parameter(torch.randn(1, n_pos, embed_dim) * .02)
        self.blocks = nn.Sequential(*[VitBlock(embed_dim, num_heads, init_values) for _ in range(depth)])
        self.norm = _RowLayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def _pos_embed(self, x):
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        if self.no_embed_class:                     # deit3: position added to the patches only
            return torch.cat((cls, x + self.pos_embed), dim=1)
        return torch.cat((cls, x), dim=1) + self.pos_embed

    def forward_features(self, x):
        return self.norm(self.blocks(self._pos_embed(self.patch_embed(x))))

    def ddp_cut(self):
        k = len(self.blocks) // 2
        return self.blocks[k - 1], [*self.blocks[k:], self.norm, self.head]

    def forward(self, x):
        return self.head(self.forward_features(x)[:, 0])


def interpolate_pos_encoding(pos_embed, new_img_size, old_img_size: int = 224, patch_size: int = 16):
    """Bicubic resize of a ViT position table for a new square resolution
    (``utils_architecture.py:22-53``; eval-time only)."""
    N = pos_embed.shape[1] - 1
    npatch = (new_img_size // patch_size) ** 2
    if npatch == N:
        return pos_embed
    cls_pos, patch_pos = pos_embed[:, 0], pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    side = int(math.sqrt(N))
    w0 = h0 = new_img_size // patch_size + 0.1      # the +0.1 of the reference (:45) avoids a floor error
    patch_pos = F.interpolate(patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2),
                              scale_factor=(w0 / side, h0 / side), mode='bicubic')
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)


# ----------------------------------------------------------------------------- arch names
def get_new_model(modelname, pretrained=True, not_original=False, updated=False, img_size=224):
    """Arch-name dispatch of ``utils_architecture.py:225-322`` for the ConvNeXt / ViT families.

    ``pretrained=True`` needs a network download in the reference; here it raises (load weights with
    ``load_state_dict`` — key names are timm's).  ``not_original=True`` swaps in the ConvStem.
    """
    if pretrained:
        raise RuntimeError("pretrained=True would download timm weights; pass pretrained=False and load a state dict")
    if modelname == 'convnext_iso':                                    # :235-239 (commented recipe)
        model = ConvNeXtIsotropic(depth=18, dim=384)
        if not_original:
            model.stem = ConvBlock(48, end_siz=8, fin_dim=432 if updated else 384)
    elif modelname == 'convnext_tiny':                                 # :241-244
        model = ConvNeXt((3, 3, 9, 3), (96, 192, 384, 768))
        if not_original:
            model.stem = ConvBlock1(48, end_siz=8)
    elif modelname == 'convnext_small':                                # :249-254
        model = ConvNeXt((3, 3, 27, 3), (96, 192, 384, 768))
        if not_original:
            model.stem = ConvBlock1(48, end_siz=8)
    elif modelname == 'convnext_base':                                 # :256-262
        model = ConvNeXt((3, 3, 27, 3), (128, 256, 512, 1024))
        if not_original:
            model.stem = ConvBlock3(64)
    elif modelname == 'convnext_large':                                # :264-269
        model = ConvNeXt((3, 3, 27, 3), (192, 384, 768, 1536))
        if not_original:
            model.stem = ConvBlock3(96)
    elif modelname in ('vit_s', 'deit_s'):                             # :271-284
        model = VisionTransformer(img_size, 16, embed_dim=384, depth=12, num_heads=6)
        if not_original:
            model.patch_embed.proj = ConvBlock(48, end_siz=8)
    elif modelname == 'vit_m':                                         # :286-291
        model = VisionTransformer(img_size, 16, embed_dim=512, depth=12, num_heads=8, init_values=1e-6,
                                  no_embed_class=True)
        if not_original:
            model.patch_embed.proj = ConvBlock2(48)
    elif modelname == 'vit_b':                                         # :297-301
        model = VisionTransformer(img_size, 16, embed_dim=768, depth=12, num_heads=12)
        if not_original:
            model.patch_embed.proj = ConvBlock(48, end_siz=16, fin_dim=None)
    else:
        # resnets / densenet / inception / *_21k go through real timm in the reference; not part of this path
        raise ValueError(f"Invalid model name {modelname!r} for the MI355X path "
                         "(supported: convnext_iso/tiny/small/base/large, vit_s, deit_s, vit_m, vit_b)")
    return model
