"""CPU oracle for the APGD adversarial-training inner loop.  TEST INFRASTRUCTURE ONLY.

This file restates, in numpy fp32, the algorithm of the reference's
``apgd_train`` (``/root/reference/autopgd_train_clean.py:123-371``).  It is the
checker the HIP path is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package (``revisiting-at_amd/``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against trajectories recorded from the reference itself
(``tests/golden/make_golden.py`` imports ``/root/reference`` in the build
container and commits inputs + expected outputs as ``.npz`` fixtures).  The
Linf path is bit-exact against those fixtures; the L2 path is checked at 1e-5
(its per-sample reductions are order dependent).

Conventions
-----------
* state tensors are ``np.float32`` arrays of shape ``[B, ...]``; every
  arithmetic operation is rounded to fp32 individually, in the association
  order the reference uses (no FMA, no re-association);
* the model is abstracted as a callable
  ``fwd_bwd(x_adv, need_grad) -> (logits[B,C], grad[B,...] | None)`` where
  ``grad = d(sum_b CE(logits_b, y_b))/d x_adv``.  ``TorchModelAdapter`` runs a
  torch module on the CPU exactly as the reference does
  (``autopgd_train_clean.py:174-192, 266-287``); ``ReplayModel`` plays back a
  recorded trajectory and verifies the iterate it is given.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# host-side schedule (autopgd_train_clean.py:152-159, 327-349)
# --------------------------------------------------------------------------
def checkpoint_schedule(n_iter: int) -> List[Tuple[int, int]]:
    """Iterations at which the step-size check fires and the window ``k`` used.

    Restates ``autopgd_train_clean.py:154-157`` (``n_iter_2``, ``n_iter_min``,
    ``size_decr``, ``k``) and the counter logic of ``:327-329, 348-349``.  The
    schedule depends on ``n_iter`` only, never on data.
    Returns ``[(i, k), ...]`` for every iteration ``i`` where ``counter3 == k``.
    """
    k = max(int(0.22 * n_iter), 1)
    k_min = max(int(0.06 * n_iter), 1)
    decr = max(int(0.03 * n_iter), 1)
    out = []
    counter = 0
    for i in range(n_iter):
        counter += 1
        if counter == k:
            out.append((i, k))
            counter = 0
            k = max(k - decr, k_min)
    return out


# --------------------------------------------------------------------------
# a2: Linf step (autopgd_train_clean.py:213-226, 260)
# --------------------------------------------------------------------------
def linf_project(t: np.ndarray, x: np.ndarray, eps: F32) -> np.ndarray:
    """``clamp(min(max(t, x-eps), x+eps), 0, 1)`` — ``:222-223``."""
    lo = x - eps
    hi = x + eps
    t = np.minimum(np.maximum(t, lo), hi)
    return np.minimum(np.maximum(t, F32(0.0)), F32(1.0))


def sign_f32(g: np.ndarray) -> np.ndarray:
    """``torch.sign``: +-1, and 0 for +-0 and NaN (``:221``)."""
    return (g > 0).astype(F32) - (g < 0).astype(F32)


def linf_step(x, x_adv, x_adv_old, grad, step_size, eps, a):
    """One Linf APGD update.  ``step_size`` is ``[B]``; ``a`` is 1.0 at i=0 else 0.75.

    Follows ``:214-226`` operation by operation.
    """
    eps = F32(eps)
    a = F32(a)
    one_minus_a = F32(1.0 - float(a))
    bshape = (-1,) + (1,) * (x.ndim - 1)
    step = step_size.reshape(bshape).astype(F32)
    grad2 = x_adv - x_adv_old                                   # :214
    x1 = x_adv + step * sign_f32(grad)                          # :221
    x1 = linf_project(x1, x, eps)                               # :222-223
    u = (x_adv + (x1 - x_adv) * a) + grad2 * one_minus_a        # :225
    return linf_project(u, x, eps)                              # :224-226


# --------------------------------------------------------------------------
# a8: L2 step (autopgd_train_clean.py:228-237, L2_norm :14-18)
# --------------------------------------------------------------------------
def l2_norm(v: np.ndarray) -> np.ndarray:
    """Per-sample ``sqrt(sum(v**2))`` in fp32, shape ``[B,1,...]`` (``:14-18``).

    The summation order differs from torch's; parity is 1e-5, not bit-exact.
    """
    b = v.shape[0]
    z = np.sqrt((v.astype(F32) ** 2).reshape(b, -1).sum(-1, dtype=F32)).astype(F32)
    return z.reshape((-1,) + (1,) * (v.ndim - 1))


def l2_project(t, x, eps):
    """``clamp(x + d/(|d|+1e-12) * min(eps, |d|), 0, 1)`` — ``:231-233``."""
    d = t - x
    n = l2_norm(d)
    r = x + d / (n + F32(1e-12)) * np.minimum(F32(eps), n)
    return np.minimum(np.maximum(r, F32(0.0)), F32(1.0))


def l2_step(x, x_adv, x_adv_old, grad, step_size, eps, a):
    a = F32(a)
    one_minus_a = F32(1.0 - float(a))
    bshape = (-1,) + (1,) * (x.ndim - 1)
    step = step_size.reshape(bshape).astype(F32)
    grad2 = x_adv - x_adv_old
    x1 = x_adv + step * grad / (l2_norm(grad) + F32(1e-12))     # :229-230
    x1 = l2_project(x1, x, eps)                                 # :231-233
    x1 = x_adv + (x1 - x_adv) * a + grad2 * one_minus_a         # :234
    return l2_project(x1, x, eps)                               # :235-237


# --------------------------------------------------------------------------
# a3: per-sample loss / prediction (criterion_dict['ce'] :113, :194-197, :291-294)
# --------------------------------------------------------------------------
def log_softmax(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.float64)
    m = z.max(-1, keepdims=True)
    s = z - m
    return s - np.log(np.exp(s).sum(-1, keepdims=True))


def ce_loss(logits: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Per-sample cross entropy, fp64 internally, rounded to fp32.

    Hard labels ``y[B]`` (``F.cross_entropy(..., reduction='none')``) or
    probability targets ``y[B,C]`` (mixup; same torch function).  Agrees with
    torch's fp32 kernel to ~1 ulp; tests use 1e-6 relative.
    """
    ls = log_softmax(logits)
    if y.ndim == 1:
        out = -ls[np.arange(ls.shape[0]), y]
    else:
        out = -(ls * y.astype(np.float64)).sum(-1)
    return out.astype(F32)


def ce_dlogits(logits: np.ndarray, y: np.ndarray) -> np.ndarray:
    """d(sum_b CE_b)/d logits in fp64: ``softmax*sum(y) - y``."""
    p = np.exp(log_softmax(logits))
    if y.ndim == 1:
        t = np.zeros_like(p)
        t[np.arange(p.shape[0]), y] = 1.0
        return p - t
    y64 = y.astype(np.float64)
    return p * y64.sum(-1, keepdims=True) - y64


def dlr_loss(logits: np.ndarray, y: np.ndarray) -> np.ndarray:
    """``dlr_loss`` — ``autopgd_train_clean.py:99-104`` (fp32, op by op)."""
    z = logits.astype(F32)
    zs = np.sort(z, axis=1)
    top = z.shape[1] - 1 - np.argmax(z[:, ::-1], axis=1)  # ind_sorted[:, -1]: torch.sort is stable -> highest index among ties
    ind = (top == y).astype(F32)
    u = np.arange(z.shape[0])
    num = z[u, y] - zs[:, -2] * ind - zs[:, -1] * (F32(1.0) - ind)
    den = zs[:, -1] - zs[:, -3] + F32(1e-12)
    return (-(num) / den).astype(F32)


def dlr_loss_targeted(logits: np.ndarray, y: np.ndarray, y_target: np.ndarray) -> np.ndarray:
    """``dlr_loss_targeted`` — ``autopgd_train_clean.py:106-111`` (fp32, op by op)."""
    z = logits.astype(F32)
    zs = np.sort(z, axis=1)
    u = np.arange(z.shape[0])
    num = z[u, y] - z[u, y_target]
    den = (zs[:, -1] - F32(0.5) * (zs[:, -3] + zs[:, -4])) + F32(1e-12)
    return (-num / den).astype(F32)


def predict(logits: np.ndarray, y: np.ndarray) -> np.ndarray:
    """``argmax(logits) == y`` (first maximal index, as torch CPU), or
    ``== argmax(y)`` for soft labels (``:194-197, 291-294``)."""
    p = np.argmax(logits, axis=1)
    if y.ndim == 1:
        return p == y
    return p == np.argmax(y, axis=1)


# --------------------------------------------------------------------------
# a6: oscillation check (check_oscillation :116-121, :329-349)
# --------------------------------------------------------------------------
def check_oscillation(loss_steps: np.ndarray, j: int, k: int, k3: float = 0.75) -> np.ndarray:
    """``t = sum_c [loss_steps[j-c] > loss_steps[j-c-1]]``; ``t <= k*k3`` as 0/1 floats.

    Negative row indices wrap exactly as Python/torch indexing does in the
    reference (row ``-1`` is the LAST row of ``loss_steps``) — the quirk noted
    in SURVEY.md §8 a6.
    """
    t = np.zeros(loss_steps.shape[1], dtype=F32)
    for c in range(k):
        t += (loss_steps[j - c] > loss_steps[j - c - 1]).astype(F32)
    return (t <= F32(k * k3)).astype(F32)


# --------------------------------------------------------------------------
# model adapters
# --------------------------------------------------------------------------
class TorchModelAdapter:
    """Runs a torch module on CPU the way the reference does (:174-192, 266-287)."""

    def __init__(self, model, y, loss: str = "ce", autocast_dtype=None):
        import torch
        self.torch = torch
        self.model = model
        self.y = torch.as_tensor(np.asarray(y))
        self.loss = loss
        self.autocast_dtype = autocast_dtype

    def __call__(self, x_adv: np.ndarray, need_grad: bool):
        torch = self.torch
        import torch.nn.functional as Fnn
        xt = torch.from_numpy(np.ascontiguousarray(x_adv))
        if need_grad:
            xt.requires_grad_()
        ctx = (torch.autocast("cpu", dtype=self.autocast_dtype)
               if self.autocast_dtype is not None else _NullCtx())
        with ctx:
            logits = self.model(xt)
            li = Fnn.cross_entropy(logits, self.y, reduction="none")
        grad = None
        if need_grad:
            grad = torch.autograd.grad(li.sum(), [xt])[0].detach().numpy()
        return logits.detach().float().numpy(), grad, li.detach().float().numpy()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def digest(a: np.ndarray) -> str:
    """sha256 of the raw fp32 bytes (+0/-0 and NaN payloads included)."""
    return hashlib.sha256(np.ascontiguousarray(a, dtype=F32).tobytes()).hexdigest()


class ReplayModel:
    """Plays back a trajectory recorded from the reference (tests/golden/*.npz).

    Call ``n`` returns ``logits[n]`` / ``grads[n]`` and records the sha256 of the
    iterate it was handed so tests can compare it with ``x_adv_sha[n]``.
    """

    def __init__(self, logits: np.ndarray, grads: np.ndarray, losses: Optional[np.ndarray] = None):
        self.logits, self.grads, self.losses = logits, grads, losses
        self.n = 0
        self.seen_sha: List[str] = []

    def __call__(self, x_adv: np.ndarray, need_grad: bool):
        n = self.n
        self.seen_sha.append(digest(x_adv))
        g = self.grads[n].copy() if need_grad else None
        li = None if self.losses is None else self.losses[n]
        self.n += 1
        return self.logits[n], g, li


# --------------------------------------------------------------------------
# the attack (autopgd_train_clean.py:123-371)
# --------------------------------------------------------------------------
@dataclass
class ApgdTrace:
    """Everything the parity tests look at besides the four return values."""
    x_adv_fed: List[np.ndarray] = field(default_factory=list)    # iterate handed to the model, per call
    loss_steps: Optional[np.ndarray] = None
    step_size: Optional[np.ndarray] = None
    n_halvings: int = 0
    flags: List[np.ndarray] = field(default_factory=list)         # per iteration uint8: 1=new best, 2=misclassified, 4=halve/restore


def apgd_train_oracle(fwd_bwd: Callable, x: np.ndarray, y: np.ndarray, norm: str, eps: float,
                      n_iter: int = 10, loss: str = "ce", soft_labels: bool = False,
                      keep_trace: bool = False, use_model_loss: bool = False,
                      y_target: Optional[np.ndarray] = None, x_init: Optional[np.ndarray] = None):
    """numpy restatement of ``apgd_train`` for ``norm in {'Linf','L2'}``.

    ``loss='dlr-targeted'`` with ``y_target`` and a start point ``x_init`` other than the clean image are the two
    things an AutoAttack-style evaluation adds on top of the reference's loop (which cannot drive them itself,
    ``:137, 181``); the update / tracking / step-size arithmetic is unchanged.

    Returns ``(x_best, acc, loss_best, x_best_adv, trace)``; the first four are
    the reference's return tuple (``:371``).  ``soft_labels`` mirrors
    ``mixup is not None``.  ``use_model_loss`` takes the per-sample loss from
    ``fwd_bwd`` (third return value) instead of recomputing it from the logits —
    used with ``ReplayModel`` to isolate the state machine bit-exactly.
    """
    if norm not in ("Linf", "L2"):
        raise ValueError("oracle covers Linf and L2 (SURVEY.md §8 a2, a8)")
    if loss not in ("ce", "dlr", "dlr-targeted"):
        raise KeyError(loss)
    if (loss == "dlr-targeted") != (y_target is not None):
        raise ValueError("y_target goes with loss='dlr-targeted'")
    x = np.asarray(x, dtype=F32)
    B = x.shape[0]
    eps32 = F32(eps)
    trace = ApgdTrace()

    def loss_of(logits, li_model):
        if use_model_loss and li_model is not None:
            return np.asarray(li_model, dtype=F32)
        if loss == "dlr-targeted":
            return dlr_loss_targeted(logits, y, y_target)
        return ce_loss(logits, y) if loss == "ce" else dlr_loss(logits, y)

    start = x if x_init is None else np.asarray(x_init, dtype=F32)
    x_adv = np.minimum(np.maximum(start.copy(), F32(0.0)), F32(1.0))  # :135, 141
    x_best = x_adv.copy()                                             # :142
    x_best_adv = x_adv.copy()                                         # :143
    loss_steps = np.zeros((n_iter, B), dtype=F32)                     # :144

    sched = dict(checkpoint_schedule(n_iter))                         # :154-157
    alpha = 2.0                                                       # :159
    step_size = np.full((B,), F32(alpha * eps), dtype=F32)            # :169-170 (python double product, then fp32)

    if keep_trace:
        trace.x_adv_fed.append(x_adv.copy())
    logits, grad, li = fwd_bwd(x_adv, True)                           # :174-192
    loss_indiv = loss_of(logits, li)
    grad_best = grad.copy()                                           # :189
    acc = predict(logits, y)                                          # :194-197
    loss_best = loss_indiv.copy()                                     # :199
    loss_best_last_check = loss_best.copy()                           # :200
    reduced_last_check = np.ones_like(loss_best)                      # :201
    x_adv_old = x_adv.copy()                                          # :205

    for i in range(n_iter):                                           # :209
        a = 0.75 if i > 0 else 1.0                                    # :218
        if norm == "Linf":
            x_new = linf_step(x, x_adv, x_adv_old, grad, step_size, eps32, a)
        else:
            x_new = l2_step(x, x_adv, x_adv_old, grad, step_size, eps32, a)
        x_adv_old = x_adv                                             # :215
        x_adv = x_new                                                 # :260
        if keep_trace:
            trace.x_adv_fed.append(x_adv.copy())

        need_grad = i < n_iter - 1                                    # :267, 281-283
        logits, g_new, li = fwd_bwd(x_adv, need_grad)                 # :273-283
        if need_grad:
            grad = g_new
        loss_indiv = loss_of(logits, li)

        pred = predict(logits, y)                                     # :291-294
        acc = np.minimum(acc, pred)                                   # :296
        ind_pred = ~pred                                              # :301
        x_best_adv = x_best_adv.copy()
        x_best_adv[ind_pred] = x_adv[ind_pred]                        # :304

        y1 = loss_indiv                                               # :319
        loss_steps[i] = y1                                            # :320
        m = y1 > loss_best                                            # :321
        x_best = x_best.copy()
        x_best[m] = x_adv[m]                                          # :322
        grad_best = grad_best.copy()
        grad_best[m] = grad[m]                                        # :323
        loss_best = loss_best.copy()
        loss_best[m] = y1[m]                                          # :324

        fl = np.zeros(B, dtype=F32)
        if i in sched:                                                # :329 (counter3 == k)
            k = sched[i]
            fl_osc = check_oscillation(loss_steps, i, k, 0.75)        # :331-332
            fl_noimp = (F32(1.0) - reduced_last_check) * (loss_best_last_check >= loss_best).astype(F32)  # :333-334
            fl = np.maximum(fl_osc, fl_noimp)                         # :335-336
            reduced_last_check = fl.copy()                            # :337
            loss_best_last_check = loss_best.copy()                   # :338
            if fl.sum() > 0:                                          # :340
                sel = fl > 0                                          # :341
                step_size = step_size.copy()
                step_size[sel] = step_size[sel] / F32(2.0)            # :342
                trace.n_halvings += int(sel.sum())
                x_adv = x_adv.copy()
                x_adv[sel] = x_best[sel]                              # :345
                grad = grad.copy()
                grad[sel] = grad_best[sel]                            # :346
        if keep_trace:
            trace.flags.append((m.astype(np.uint8) | (ind_pred.astype(np.uint8) << 1)
                                | ((fl > 0).astype(np.uint8) << 2)))

    trace.loss_steps = loss_steps
    trace.step_size = step_size
    return x_best, acc, loss_best, x_best_adv, trace


def check_imgs(adv: np.ndarray, x: np.ndarray, norm: str, eps: float):
    """Invariants of ``utils_eval.py:67-81``: per-sample perturbation norm, NaNs, range.

    Returns ``(max_norm, n_nan, lo, hi)``; callers assert ``max_norm <= eps`` (+1 ulp
    for the fp32 ``x+eps`` rounding), ``n_nan == 0`` and ``0 <= lo, hi <= 1``.
    """
    d = (adv.astype(np.float64) - x.astype(np.float64)).reshape(x.shape[0], -1)
    if norm == "Linf":
        r = np.abs(d).max(1)
    elif norm == "L2":
        r = np.sqrt((d ** 2).sum(1))
    else:
        r = np.abs(d).sum(1)
    return float(r.max()), int(np.isnan(adv).sum()), float(adv.min()), float(adv.max())
