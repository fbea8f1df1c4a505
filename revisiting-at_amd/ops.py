"""Model-side operators of the hot path.

Each operator has ONE product implementation.  Where a hand-written gfx950 kernel exists
(``csrc/model_kernels.hip``) it is used unconditionally for device tensors and raises if the
library is missing; plain convolutions / GEMMs are PyTorch-ROCm library calls (MIOpen,
hipBLASLt).  ``APGD_OPS=eager`` switches the fused operators back to their library-call
composition for A/B timing only (bench.py records which mode ran).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

MODE = os.environ.get("APGD_OPS", "hip")


def layer_norm_cf(x, weight, bias, eps):
    """LayerNorm over dim 1 of ``[N,C,H,W]`` (``utils_architecture.py:76-81``; timm LayerNorm2d)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), weight.shape, weight, bias, eps).permute(0, 3, 1, 2)


def layer_norm_cf_gelu(x, weight, bias, eps):
    """``GELU(LN_cf(x))`` — the ConvStem pair (``utils_architecture.py:128-129`` etc.)."""
    return F.gelu(layer_norm_cf(x, weight, bias, eps))


def convnext_block(x, dw_w, dw_b, ln_w, ln_b, eps, w1, b1, w2, b2, gamma):
    """``x + gamma * fc2(GELU(fc1(LN(dw7x7(x)))))`` on ``[N,C,H,W]`` (``models/convnext.py:37-50``)."""
    y = F.conv2d(x, dw_w, dw_b, padding=3, groups=x.shape[1]).permute(0, 2, 3, 1)
    y = F.layer_norm(y, ln_w.shape, ln_w, ln_b, eps)
    y = F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    if gamma is not None:
        y = y * gamma
    return x + y.permute(0, 3, 1, 2)


def attention(qkv, num_heads, scale):
    """Multi-head softmax attention from a packed ``[B,N,3C]`` projection -> ``[B,N,C]``
    (timm 0.8 ``Attention.forward``)."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).unbind(0)
    a = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, C)
