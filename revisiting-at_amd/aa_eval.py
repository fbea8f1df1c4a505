"""APGD-CE + APGD-T robustness evaluation on the MI355X attack kernels — the path ``AA_eval.py`` drives
(``/root/reference/AA_eval.py:226-239``: ``AutoAttack(model, norm, eps, version='standard')`` with
``attacks_to_run = ['apgd-ce', 'apgd-t']`` and ``run_standard_evaluation(x, y, bs)``).

The attack arithmetic of that path lives in the third-party ``autoattack`` 0.1 package, whose source is not in the
reference tree (**parity unpinned**).  What IS in the tree and pins this module:
the APGD update / tracking / step-size schedule (``autopgd_train_clean.py:123-371``, a trimmed copy of AutoAttack's
``autopgd_base.py``; bit-exact fixtures incl. ``n_iter=100``), ``dlr_loss_targeted`` (``:106-111``) and the caller
shape.  On top of the shared device loop (``apgd._apgd_core``) this adds what AutoAttack's 'standard' APGD does
differently from ``apgd_train`` (Appendix C):

  * random start inside the eps-ball: ``x + eps * t / max|t|``, ``t ~ U(-1, 1)`` per sample (Linf);
  * 100 iterations, one restart; APGD-T: 9 target classes, ``y_target`` = the (t+1)-th most likely class on the
    clean image, loss = targeted DLR;
  * only still-robust points are attacked (clean-correct, not yet broken by an earlier attack);
  * the returned adversarial is the misclassified iterate (``x_best_adv`` semantics).

Multi-GPU (BASELINE config #5): the evaluation set is sharded over ranks (``rank::world``), the attack runs without
any collective, and the robust / clean counts are summed once at the end (``robust_accuracy``).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from .apgd import _apgd_core

LOSS_KIND = {'ce': 0, 'dlr': 1, 'dlr-targeted': 2}
EPS_DICT = {'imagenet': {'Linf': 4. / 255., 'L2': 2., 'L1': 75.}}          # AA_eval.py:33


def random_start(x: torch.Tensor, eps: float, norm: str = 'Linf', generator: Optional[torch.Generator] = None):
    """AutoAttack's APGD start point for Linf: ``x + eps * t / max_i |t_i|`` with ``t ~ U(-1, 1)`` per sample."""
    if norm != 'Linf':
        raise NotImplementedError("random start is implemented for Linf (the norm AA_eval.py evaluates by default)")
    t = 2 * torch.rand(x.shape, device=x.device, dtype=torch.float32, generator=generator) - 1
    scale = t.reshape(t.shape[0], -1).abs().amax(dim=1).clamp_min(1e-12).view(-1, *([1] * (x.dim() - 1)))
    return x + eps * t / scale


def apgd_attack(model, x, y, norm='Linf', eps=4. / 255., n_iter=100, loss='ce', y_target=None, use_rs=True,
                generator: Optional[torch.Generator] = None, graph: bool = False, n_real: Optional[int] = None):
    """One APGD run (one restart) -> ``(x_best_adv, acc, loss_best, x_best)``.

    ``acc[b]`` is True iff sample ``b`` was classified correctly at the start point and at every iterate (it is still
    robust); ``x_best_adv[b]`` is the last misclassified iterate (the start point if none).  ``loss='dlr-targeted'``
    needs ``y_target`` (int64 ``[B]``).
    """
    assert not model.training
    if loss not in LOSS_KIND:
        raise KeyError(loss)
    kind = LOSS_KIND[loss]
    if (kind == 2) != (y_target is not None):
        raise ValueError("y_target goes with loss='dlr-targeted'")
    if norm not in ('Linf', 'L2'):
        raise NotImplementedError(f"norm={norm!r}")
    # ``n_real`` (run_standard_evaluation's padded batches): rows n_real .. are copies of row 0 that only fill the batch up to a
    # size already met (and captured); the random start is drawn for the real rows only, so the random stream - and with it every
    # real row's trajectory - is that of the unpadded call
    x_init = None
    if use_rs:
        if n_real is not None and n_real < x.shape[0]:
            x_init = torch.cat([random_start(x[:n_real], eps, norm, generator), x[n_real:]], 0)
        else:
            x_init = random_start(x, eps, norm, generator)
    if graph:
        from . import graphed
        x_best, acc, loss_best, x_best_adv = graphed.run(model, x, y, norm, eps, n_iter, kind, False, y_target=y_target, x_init=x_init)
    else:
        x_best, acc, loss_best, x_best_adv = _apgd_core(model, x, y, norm, eps, n_iter, kind, soft=False,
                                                        y_target=y_target, x_init=x_init)
    return x_best_adv, acc, loss_best, x_best


def _bucket(n: int, bs: int) -> int:
    """The padded size of an ``n``-row attack batch: the smallest of ``bs, ceil(bs/2), ceil(bs/4), ... >= 8`` that holds it."""
    size = max(bs, n)
    while True:
        half = (size + 1) // 2
        if half < n or half < 8:
            return size
        size = half


@torch.no_grad()
def _predict(model, x, bs):
    out = []
    for i in range(0, x.shape[0], bs):
        out.append(model(x[i:i + bs]).float())
    return torch.cat(out) if out else torch.empty(0, 0, device=x.device)


def run_standard_evaluation(model, x, y, bs=200, norm='Linf', eps=4. / 255.,
                            attacks_to_run: Sequence[str] = ('apgd-ce', 'apgd-t'), n_iter=100, n_target_classes=9,
                            seed=0, rank=0, world=1, device=None, amp_dtype=None, verbose=False, buckets: bool = False,
                            graph: bool = False) -> Tuple[torch.Tensor, dict]:
    """``AutoAttack.run_standard_evaluation`` for ``attacks_to_run ⊆ {'apgd-ce', 'apgd-t'}`` (AA_eval.py:230-239).

    ``x`` [n,3,H,W] fp32 in [0,1] and ``y`` [n] may live on the host (as in AA_eval.py:116); this rank evaluates
    samples ``rank::world`` in batches of ``bs`` on ``device``.  Returns ``(x_adv_shard, stats)`` with
    ``stats = {'n', 'clean_correct', 'robust', 'attack_runs', 'sample_iters'}`` (counts on this rank; see ``robust_accuracy``;
    the last two say how much work the evaluation was: APGD runs started, and samples x iterations they covered).

    ``buckets``: the still-robust subset an attack runs on has an arbitrary size - every one a batch shape the libraries (and a
    captured program) have not met.  With ``buckets`` it is padded to the next of a few fixed sizes (``bs``, then halves down to
    8) with copies of its first row; the padding rows are never read back (samples are independent: eval mode, LayerNorm models)
    and draw no random numbers.  ``graph``: the attack runs replay from hipGraphs (``graphed.run``: captured on the third call
    with a shape, which the buckets make recur).

    Both are THROUGHPUT options and a tolerance, not an identity: a padded (or replayed) run shows the GEMM / convolution kernels
    another batch shape - other tiles and split-K choices, under autocast ``cnx_gemm_nt`` and two streams - so a real row's logits
    agree with the unpadded eager run's to rounding (fp32: ~1e-6 relative), not bit for bit, and a sample whose margin sits inside
    that rounding can end on the other side of robust / non-robust.  ``buckets=False, graph=False`` (the defaults) is the
    evaluation whose attack runs replay through the oracle bit for bit (tests/test_gpu_configs.py) and the one to report robust
    accuracy from; ``tests/test_gpu_apgd.py::test_run_standard_evaluation_with_padded_batches_and_graph_replay`` bounds the
    difference between the two on a fixed seed (robust masks, counts).
    """
    assert not model.training
    for a in attacks_to_run:
        if a not in ('apgd-ce', 'apgd-t'):
            raise NotImplementedError(f"attack {a!r}: the MI355X path implements apgd-ce and apgd-t")
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    if device.type != 'cuda':
        raise _lib.ApgdHipError("the evaluation attacks need the MI355X; there is no CPU fallback")
    xs, ys = x[rank::world], y[rank::world]
    n = xs.shape[0]
    x_adv_all = xs.clone()
    gen = torch.Generator(device=device).manual_seed(seed * 1000003 + rank)
    clean_correct = robust_total = attack_runs = sample_iters = 0
    ctx = torch.autocast('cuda', dtype=amp_dtype) if amp_dtype is not None else torch.autocast('cuda', enabled=False)
    for b0 in range(0, n, bs):
        xb = xs[b0:b0 + bs].to(device, non_blocking=True).float().contiguous()
        yb = ys[b0:b0 + bs].to(device, non_blocking=True).long()
        with ctx:
            logits = _predict(model, xb, bs)
            robust = logits.argmax(1) == yb                                  # clean accuracy first (AA_eval.py:192)
            clean_correct += int(robust.sum())
            x_adv = xb.clone()
            # target classes come from the clean logits: the (t+1)-th most likely class, t = 1..n_target_classes
            order = logits.argsort(dim=1, descending=True)
            for attack in attacks_to_run:
                targets = [None] if attack == 'apgd-ce' else list(range(2, n_target_classes + 2))
                for tc in targets:
                    idx = robust.nonzero().squeeze(1)                        # only still-robust points are attacked
                    if idx.numel() == 0:
                        break
                    n_real = int(idx.numel())
                    attack_runs += 1
                    sample_iters += n_real * n_iter
                    if tc is not None and tc > logits.shape[1]:
                        break
                    idp = idx
                    if buckets:
                        nb = _bucket(n_real, bs)
                        if nb > n_real:
                            idp = torch.cat([idx, idx[:1].expand(nb - n_real)])
                    xi, yi = xb[idp].contiguous(), yb[idp]
                    if tc is None:
                        xa, acc, _, _ = apgd_attack(model, xi, yi, norm, eps, n_iter, 'ce', None, True, gen, graph, n_real)
                    else:
                        yt = order[idp, tc - 1].contiguous()
                        xa, acc, _, _ = apgd_attack(model, xi, yi, norm, eps, n_iter, 'dlr-targeted', yt, True, gen, graph, n_real)
                    xa, acc = xa[:n_real], acc[:n_real]
                    broken = ~acc
                    if broken.any():
                        x_adv[idx[broken]] = xa[broken]
                        robust[idx[broken]] = False
                    if verbose:
                        print(f"[rank {rank}] batch {b0 // bs} {attack}{'' if tc is None else f' t={tc}'}: "
                              f"robust {int(robust.sum())}/{xb.shape[0]}")
        robust_total += int(robust.sum())
        x_adv_all[b0:b0 + bs] = x_adv.to(x_adv_all.device)
    return x_adv_all, {'n': n, 'clean_correct': clean_correct, 'robust': robust_total, 'attack_runs': attack_runs,
                       'sample_iters': sample_iters}


def robust_accuracy(stats: dict, device=None) -> Tuple[float, float]:
    """(clean accuracy, robust accuracy) over all ranks: the only exchange of the evaluation — one sum of three counts
    (RCCL all-reduce when ``torch.distributed`` is initialised, otherwise this rank's own numbers)."""
    import torch.distributed as dist
    t = torch.tensor([stats['n'], stats['clean_correct'], stats['robust']], dtype=torch.float64,
                     device=device if device is not None else 'cpu')
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t)
    n, c, r = t.tolist()
    return (c / n if n else 0.0), (r / n if n else 0.0)
