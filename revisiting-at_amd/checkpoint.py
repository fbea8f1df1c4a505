"""Checkpoint key handling of the reference (SURVEY.md §5 "Checkpoint / resume", §8f-4): plain ``torch.save`` dicts with
timm key names, prefixed by the wrappers (``module.`` DDP, ``base_model.`` WrappedModel, ``model.`` normalize_model).

* ``load_weights`` — the three-way fallback of ``main.py:856-871`` (always strip ``module.``; as-is → add ``base_model.`` →
  strip ``base_model.``), so weights saved from a wrapped, DDP or bare model load into any of the three.
* ``clean_eval_keys`` — the evaluation-side cleanup of ``AA_eval.py:186-188``.
* ``save_weights`` / ``save_full`` — the files ``main.py:737-756`` writes (``weights_{epoch}.pt``, ``weights_ema_{epoch}.pt``,
  ``full_model_{epoch}.pth`` every 5 epochs; no GradScaler state on the bf16 path).
"""
from __future__ import annotations

import os
from typing import Dict

import torch


def strip_module(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k.replace('module.', ''): v for k, v in sd.items()}                       # main.py:858


def clean_eval_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    sd = {k.replace('module.', ''): v for k, v in sd.items()}                         # AA_eval.py:186
    sd = {k.replace('base_model.', ''): v for k, v in sd.items()}                     # :187
    return {k.replace('se_', 'se_module.'): v for k, v in sd.items()}                 # :188


def load_weights(model: torch.nn.Module, ckpt) -> str:
    """``ckpt``: path or state dict.  Returns which of the reference's three attempts succeeded."""
    if isinstance(ckpt, (str, os.PathLike)):
        ckpt = torch.load(ckpt, map_location='cpu')
    if isinstance(ckpt.get('model_state_dict', None), dict):              # a full_model_{epoch}.pth file (main.py:742)
        ckpt = ckpt['model_state_dict']
    sd = strip_module(ckpt)
    try:
        model.load_state_dict(sd)
        return 'standard loading'                                                     # main.py:860-861
    except RuntimeError:
        try:
            model.load_state_dict({f'base_model.{k}': v for k, v in sd.items()})
            return 'loaded from clean model'                                          # :864-867
        except RuntimeError:
            model.load_state_dict({k.replace('base_model.', ''): v for k, v in sd.items()})
            return 'loaded'                                                           # :868-871


def save_weights(model, folder, epoch: int, ema_state: Dict[str, torch.Tensor] = None):
    """``model``: anything with ``state_dict()`` - pass the ``ATTrainStep`` itself under N > 1 ranks: its ``state_dict()`` carries the
    ``module.`` prefix of the reference's DDP-wrapped model also on the flat gradient path (no DDP object there)."""
    os.makedirs(folder, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(folder, f'weights_{epoch}.pt'))       # main.py:738
    if ema_state is not None:
        torch.save(ema_state, os.path.join(folder, f'weights_ema_{epoch}.pt'))        # :740


def save_full(model, optimizer, folder, epoch: int, ema_state=None):
    os.makedirs(folder, exist_ok=True)
    d = {'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(), 'epoch': epoch}
    if ema_state is not None:
        d['state_dict_ema'] = ema_state                                               # :747
    torch.save(d, os.path.join(folder, f'full_model_{epoch}.pth'))                    # :742-756
