"""TEST INFRASTRUCTURE - numpy-fp32 restatement of the L1 branch of the reference's ``apgd_train``
(``/root/reference/autopgd_train_clean.py``: parameters ``:160-167``, sparse signed step ``:239-250``, ``L1_projection`` ``:24-91``,
sparsity adaptation ``:351-362``; everything else - forward / loss / tracking - as in ``apgd_oracle``).  SURVEY.md marks the branch
"next": it is not on the AT path the benchmark measures (and the reference's own code takes NCHW-contiguous inputs only, ``:240``).

Pinned by ``tests/golden/apgd_l1_*.npz`` (recorded from the reference).  Row sums and prefix sums over thousands of fp32 terms are
order-dependent (torch's vectorised sum vs numpy's pairwise one), so with the recorded model outputs replayed the iterates agree with
the reference to ~1e-6 relative, not bit for bit - the same status as the L2 branch.
"""
from __future__ import annotations

import math
from typing import Callable

import numpy as np

from .apgd_oracle import ApgdTrace, F32, ce_loss, predict, sign_f32


def l1_projection(x2: np.ndarray, y2: np.ndarray, eps1: float) -> np.ndarray:
    """``L1_projection`` (``:24-91``): delta such that ``||y2 + delta||_1 <= eps1`` and ``0 <= x2 + y2 + delta <= 1``, as the
    reference computes it: sort of the 2n break points, prefix sums, a bisection per sample over the sorted positions."""
    B = x2.shape[0]
    x = x2.reshape(B, -1).astype(F32)
    y = y2.reshape(B, -1).astype(F32)
    n = x.shape[1]
    sigma = sign_f32(y)
    u = np.minimum((F32(1.0) - x) - y, x + y)
    u = np.minimum(F32(0.0), u)
    l = -np.abs(y)
    d = u.copy()
    neg = -np.concatenate((u, l), 1)
    indbs = np.argsort(neg, axis=1, kind="stable")
    bs = np.take_along_axis(neg, indbs, 1)
    bs2 = np.concatenate((bs[:, 1:], np.zeros((B, 1), F32)), 1)
    inu = F32(2.0) * (indbs < n).astype(F32) - F32(1.0)
    size1 = np.cumsum(inu, axis=1, dtype=F32)
    s1 = -u.sum(1, dtype=F32)
    c = F32(eps1) - np.abs(y).sum(1, dtype=F32)
    c5 = (s1 + c) < 0
    c2 = np.nonzero(c5)[0]
    s = s1[:, None] + np.cumsum((bs2 - bs) * size1, axis=1, dtype=F32)
    if len(c2):
        lb = np.zeros(len(c2), F32)
        ub = np.full(len(c2), F32(2 * n - 1), F32)
        for _ in range(int(math.ceil(math.log2(2 * n)))):
            mid = np.floor((lb + ub) / F32(2.0))
            mi = mid.astype(np.int64)
            c8 = (s[c2, mi] + c[c2]) < 0
            lb = np.where(c8, mid, lb)
            ub = np.where(c8, ub, mid)
        lb2 = lb.astype(np.int64)
        alpha = (-s[c2, lb2] - c[c2]) / size1[c2, lb2 + 1] + bs2[c2, lb2]
        d[c2] = -np.minimum(np.maximum(-u[c2], alpha[:, None]), -l[c2])
    return (sigma * d).reshape(x2.shape).astype(F32)


def l1_step(x, x_adv, grad, step_size, topk, eps):
    """``:239-250``: keep the largest ``topk`` share of |grad| per sample, step ``step_size / #kept`` along its sign, project."""
    B = x.shape[0]
    n_fts = int(np.prod(x.shape[1:]))
    g = grad.reshape(B, -1).astype(F32)
    srt = np.sort(np.abs(g), axis=1)
    pos = np.clip((F32(1.0) - topk) * F32(n_fts), 0, n_fts - 1).astype(np.int64)       # .long(): truncation of a non-negative float
    thr = srt[np.arange(B), pos]
    sparse = g * (np.abs(g) >= thr[:, None]).astype(F32)
    sg = sign_f32(sparse)
    cnt = np.abs(sg).sum(1, dtype=F32) + F32(1e-10)
    shp = (B,) + (1,) * (x.ndim - 1)
    x1 = x_adv + (step_size.reshape(shp) * sg.reshape(x.shape)) / cnt.reshape(shp)
    delta_u = x1 - x
    delta_p = l1_projection(x, delta_u, eps)
    return ((x + delta_u) + delta_p).astype(F32)


def apgd_train_l1_oracle(fwd_bwd: Callable, x: np.ndarray, y: np.ndarray, eps: float, n_iter: int = 10, is_train: bool = True,
                         keep_trace: bool = False):
    """``apgd_train(norm='L1', loss='ce')``.  Returns ``(x_best, acc, loss_best, x_best_adv, trace)``."""
    x = np.asarray(x, dtype=F32)
    B = x.shape[0]
    n_fts = int(np.prod(x.shape[1:]))
    trace = ApgdTrace()
    x_adv = np.minimum(np.maximum(x.copy(), F32(0.0)), F32(1.0))       # :135, 141
    x_best, x_best_adv = x_adv.copy(), x_adv.copy()
    loss_steps = np.zeros((n_iter, B), dtype=F32)
    k = max(int(.04 * n_iter), 1)                                      # :161
    topk = np.full(B, F32(.05 if is_train else .2), F32)               # :162-163
    sp_old = np.full(B, F32(n_fts), F32)                               # :164
    adasp_redstep, adasp_minstep, alpha = 1.5, 10., 1.                 # :165-167
    step_size = np.full(B, F32(alpha * eps), F32)                      # :169-170
    counter3 = 0

    if keep_trace:
        trace.x_adv_fed.append(x_adv.copy())
    logits, grad, _ = fwd_bwd(x_adv, True)                             # :174-192
    grad = np.asarray(grad, F32)
    grad_best = grad.copy()
    acc = predict(logits, y)
    loss_best = ce_loss(logits, y)

    for i in range(n_iter):                                            # :209
        x_adv = l1_step(x, x_adv, grad, step_size, topk, eps)          # :239-250, 260
        if keep_trace:
            trace.x_adv_fed.append(x_adv.copy())
        need_grad = i < n_iter - 1
        logits, g_new, _ = fwd_bwd(x_adv, need_grad)
        if need_grad:
            grad = np.asarray(g_new, F32)
        loss_indiv = ce_loss(logits, y)
        pred = predict(logits, y)
        acc = np.minimum(acc, pred)
        x_best_adv = x_best_adv.copy()
        x_best_adv[~pred] = x_adv[~pred]
        loss_steps[i] = loss_indiv
        m = loss_indiv > loss_best
        x_best = x_best.copy(); x_best[m] = x_adv[m]
        grad_best = grad_best.copy(); grad_best[m] = grad[m]
        loss_best = loss_best.copy(); loss_best[m] = loss_indiv[m]
        counter3 += 1
        if counter3 == k:                                              # :329, 351-362: adapt the sparsity
            sp_curr = ((x_best - x) != 0).reshape(B, -1).sum(1).astype(F32)          # L0_norm, :20-21 (integer count -> float division)
            fl_redtopk = (sp_curr / sp_old) < F32(.95)
            topk = ((sp_curr / F32(n_fts)) / F32(1.5)).astype(F32)
            step_size = step_size.copy()
            step_size[fl_redtopk] = F32(alpha * eps)
            step_size[~fl_redtopk] = step_size[~fl_redtopk] / F32(adasp_redstep)
            step_size = np.clip(step_size, F32(alpha * eps / adasp_minstep), F32(alpha * eps)).astype(F32)
            sp_old = sp_curr.copy()
            x_adv = x_adv.copy(); x_adv[fl_redtopk] = x_best[fl_redtopk]
            grad = grad.copy(); grad[fl_redtopk] = grad_best[fl_redtopk]
            counter3 = 0
    trace.loss_steps = loss_steps
    trace.step_size = step_size
    return x_best, acc, loss_best, x_best_adv, trace
