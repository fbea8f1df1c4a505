"""Attack wiring of the reference trainer (``/root/reference/main.py:175-184, 831-844``).

Keeps the dotted ``adv.*`` flag names so a launch line such as
``--adv.attack apgd --adv.n_iter 2 --adv.norm Linf --adv.eps 0.0157`` (``run_train.sh:10-18``)
maps one to one.  fastargs is not a dependency: this is a 40-line parser for the ``adv``
section only; the rest of the reference's configuration is out of scope.
"""
from __future__ import annotations

import functools
from dataclasses import dataclass, fields
from typing import Optional, Sequence

from .apgd import apgd_train
from .fgsm import fgsm_train
from .wrapped_model import WrappedModel


@dataclass
class AdvConfig:
    """``Section('adv', 'adversarial training options')`` — defaults of ``main.py:175-184``."""
    attack: str = 'none'          # 'none' | 'apgd' | 'fgsm'
    norm: str = 'Linf'
    eps: float = 4. / 255.
    n_iter: int = 2
    verbose: int = 0
    noise_level: float = 1.
    skip_projection: int = 0
    alpha: float = 1.
    graph: int = 0                # (addition of this path) 1: replay the attack from hipGraphs - apgd_train(graph=True), graphed.py

    @classmethod
    def from_argv(cls, argv: Sequence[str]) -> "AdvConfig":
        """Parse ``--adv.<name> <value>`` pairs; other flags are ignored (they belong to the trainer)."""
        cfg = cls()
        types = {f.name: f.type for f in fields(cls)}
        i = 0
        argv = list(argv)
        while i < len(argv):
            tok = argv[i]
            if tok.startswith('--adv.'):
                name, _, val = tok[len('--adv.'):].partition('=')
                if not val:
                    i += 1
                    if i >= len(argv):
                        raise ValueError(f"flag {tok} needs a value")
                    val = argv[i]
                if name not in types:
                    raise ValueError(f"unknown flag --adv.{name}")
                cur = getattr(cfg, name)
                if isinstance(cur, float) and '/' in val:      # e.g. --adv.eps 4/255
                    num, _, den = val.partition('/')
                    setattr(cfg, name, float(num) / float(den))
                else:
                    setattr(cfg, name, type(cur)(val))
            i += 1
        return cfg


def build_perturb(cfg: AdvConfig, mixup=None):
    """``functools.partial(apgd_train, norm, eps, n_iter, verbose, mixup)`` (``main.py:834-835``) or
    ``functools.partial(fgsm_train, eps, use_rs=True, alpha, noise_level, skip_projection)`` (``main.py:836-842``).

    Returns None for ``adv.attack == 'none'`` (the model is then not wrapped, ``main.py:831``).
    """
    if cfg.attack == 'none':
        return None
    if cfg.attack == 'apgd':
        return functools.partial(apgd_train, norm=cfg.norm, eps=cfg.eps, n_iter=cfg.n_iter,
                                 verbose=cfg.verbose == 1, mixup=mixup, **({"graph": True} if cfg.graph else {}))
    if cfg.attack == 'fgsm':                                  # main.py:836-842
        return functools.partial(fgsm_train, eps=cfg.eps, use_rs=True, alpha=cfg.alpha, noise_level=cfg.noise_level,
                                 skip_projection=cfg.skip_projection == 1)
    raise ValueError(f"unknown adv.attack {cfg.attack!r}")


def wrap_model_for_at(model, cfg: AdvConfig, mixup=None):
    """``model = WrappedModel(model, perturb, verbose)`` when an attack is configured (``main.py:831-844``)."""
    perturb = build_perturb(cfg, mixup)
    if perturb is None:
        return model
    return WrappedModel(model, perturb, verbose=cfg.verbose == 1)
