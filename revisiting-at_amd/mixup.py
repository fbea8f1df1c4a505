"""Mixup / CutMix with soft targets on the device — the step immediately before the hot path
(``/root/reference/main.py:599-607, 965-966``; parameters from ``parserr.py:17-41``: mixup 0.8, cutmix 1.0,
prob 1.0, switch 0.5, mode 'batch', label smoothing ``training.label_smoothing``) and the loss it pairs with
(``main.py:461-466``: ``timm.loss.SoftTargetCrossEntropy``).

The reference takes both from timm 0.8 (not in its tree: **parity unpinned**); this is a restatement of
timm's *batch* mode: one ``lam`` per batch, partner sample = the batch flipped, CutMix box with the area-corrected
``lam``, smoothed one-hot targets mixed with the same ``lam``.  What matters to the attack kernels is the layout it
defines: ``y`` becomes fp32 ``[B, n_cls]`` probabilities (``autopgd_train_clean.py:194-197, 291-294`` then compare
``argmax(logits)`` with ``argmax(y)``).  ``lam`` and the box are drawn on the host (as timm does, from numpy); every
tensor operation runs on the tensors' device with no synchronisation.
"""
from __future__ import annotations

import numpy as np
import torch


def one_hot(target, num_classes, on_value=1.0, off_value=0.0):
    t = target.long().view(-1, 1)
    return torch.full((t.shape[0], num_classes), off_value, device=t.device, dtype=torch.float32).scatter_(1, t, on_value)


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0):
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    y1 = one_hot(target, num_classes, on, off)
    y2 = one_hot(target.flip(0), num_classes, on, off)
    return y1 * lam + y2 * (1.0 - lam)


def rand_bbox(img_shape, lam, rng, margin=0.0):
    """CutMix box of area ratio ``1 - lam`` centred uniformly (timm ``rand_bbox``); returns (yl, yh, xl, xh)."""
    H, W = img_shape[-2:]
    ratio = np.sqrt(1.0 - lam)
    cut_h, cut_w = int(H * ratio), int(W * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = rng.randint(0 + margin_y, H - margin_y)
    cx = rng.randint(0 + margin_x, W - margin_x)
    yl, yh = np.clip(cy - cut_h // 2, 0, H), np.clip(cy + cut_h // 2, 0, H)
    xl, xh = np.clip(cx - cut_w // 2, 0, W), np.clip(cx + cut_w // 2, 0, W)
    return int(yl), int(yh), int(xl), int(xh)


class Mixup:
    """``timm.data.Mixup`` (batch mode) as the reference constructs it (``main.py:604-607``)."""

    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000, seed=None):
        if cutmix_minmax is not None:
            raise NotImplementedError("cutmix_minmax is None in the reference's presets (parserr.py:29)")
        if mode != 'batch':
            raise NotImplementedError("the reference uses mixup_mode='batch' (parserr.py:32)")
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.correct_lam = correct_lam
        self.mixup_enabled = True
        self.rng = np.random.RandomState(seed)

    def _params(self):
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and self.rng.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = self.rng.rand() < self.switch_prob
                alpha = self.cutmix_alpha if use_cutmix else self.mixup_alpha
            elif self.mixup_alpha > 0.0:
                alpha = self.mixup_alpha
            elif self.cutmix_alpha > 0.0:
                use_cutmix, alpha = True, self.cutmix_alpha
            else:
                raise ValueError("one of mixup_alpha > 0, cutmix_alpha > 0 must be set")
            lam = float(self.rng.beta(alpha, alpha))
        return lam, use_cutmix

    def __call__(self, x, target):
        if x.shape[0] % 2 != 0:
            raise ValueError("Batch size should be even when using this")
        lam, use_cutmix = self._params()
        if lam != 1.0:
            if use_cutmix:
                yl, yh, xl, xh = rand_bbox(x.shape, lam, self.rng)
                if self.correct_lam:
                    lam = 1.0 - (yh - yl) * (xh - xl) / float(x.shape[-2] * x.shape[-1])
                x = x.clone()
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
            else:
                x = x * lam + x.flip(0) * (1.0 - lam)
        return x, mixup_target(target, self.num_classes, lam, self.label_smoothing)


class SoftTargetCrossEntropy(torch.nn.Module):
    """``timm.loss.SoftTargetCrossEntropy``: ``mean_b sum_c -target * log_softmax(x)`` (``main.py:466``)."""

    def forward(self, x, target):
        return torch.sum(-target * torch.log_softmax(x.float(), dim=-1), dim=-1).mean()
