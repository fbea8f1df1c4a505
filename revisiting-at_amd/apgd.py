"""MI355X-native APGD inner loop behind the reference's ``apgd_train`` boundary.

Host mirror of ``/root/reference/autopgd_train_clean.py:123-371``: same name, same
arguments, same return tuple, same error behaviour — but the per-iteration element-wise
update, the per-sample loss / prediction, the best-point tracking and the step-size state
machine run as hand-written HIP kernels (``csrc/apgd_kernels.hip`` through the C ABI of
``include/apgd_hip.h``) and the loop performs **no host synchronisation**: the check-point
schedule depends only on ``n_iter`` and is computed on the host, every data-dependent
decision stays on the device as a per-sample flag byte.

What the model sees is unchanged: ``n_iter + 1`` eval-mode forwards and ``n_iter``
input-gradient backwards, issued from Python on the current torch stream.  The gradient
of ``sum_b CE_b`` w.r.t. the logits is produced by the loss kernel and fed to
``torch.autograd.grad(logits, x_adv, grad_outputs=dlogits)``, so no parameter gradient is
touched (``autopgd_train_clean.py:182-185``).

There is no CPU fallback: CPU tensors or a missing ``libapgd_hip.so`` raise.
"""
from __future__ import annotations

import contextlib
import weakref
import math
import os
from typing import List, Optional, Tuple

import torch

from . import _lib, ops

__all__ = ["apgd_train", "checkpoint_schedule", "criterion_names", "ApgdWorkspace"]

# bench.py sets this to a list to have every Linf-update launch bracketed by HIP events recorded on
# the stream the kernel runs on: entries are (name, iteration, start_event, end_event, gradient element bytes).
PROFILE_EVENTS = None

# launch shape of the general Linf update (blocks per sample, unroll, non-temporal loads) handed to apgd_linf_step_f32_ex; None = the
# library's default (tools/k1_instep.py sweeps it inside the AT step, where the kernel's operands come from HBM, not from the cache)
K1_SHAPE = None

# Linf only reads sign(grad) (:221): let our own stem kernel hand the attack int8 signs instead of the fp32 gradient
# (ops.grad_sign_sink).  Same decisions bit for bit; 17 instead of 20 bytes per element in the update kernel.
USE_SIGN_SINK = True

# the Linf update kernel also performs the row moves of the iteration before it (apgd_linf_step_track_f32): one element-wise
# pass per iteration instead of two.  False = the separate apgd_track_rows pass of rounds 1-4 (kept for A/B and the tests)
FUSED_TRACKING = os.environ.get("APGD_FUSED_TRACKING", "1") not in ("0", "")

# default of apgd_train(graph=None): hipGraph replay of the attack (graphed.py)
GRAPH_DEFAULT = os.environ.get("APGD_GRAPH", "0") not in ("0", "")

# models whose first backward through the sink showed a second consumer of the attack iterate (see _model_fwd_bwd); weak: the id of
# a dead model may be handed to a new one, which must get the sink again
_SINK_REFUSED = weakref.WeakSet()


def _sink_refused(model) -> bool:
    try:
        return model in _SINK_REFUSED
    except TypeError:                                        # not weak-referenceable: never recorded
        return False

# losses the reference's criterion_dict names (autopgd_train_clean.py:113-114)
criterion_names = ("ce", "softloss", "dlr", "dlr-targeted")


def checkpoint_schedule(n_iter: int) -> List[Tuple[int, int]]:
    """``[(i, k)]``: iterations at which ``counter3 == k`` fires and the window ``k`` used.

    ``autopgd_train_clean.py:154-157`` (n_iter_2 / n_iter_min / size_decr) and ``:327-329,
    348-349``.  Depends on ``n_iter`` only, which is what lets the loop run without reading
    anything back from the device.
    """
    k = max(int(0.22 * n_iter), 1)
    k_min = max(int(0.06 * n_iter), 1)
    size_decr = max(int(0.03 * n_iter), 1)
    out, counter = [], 0
    for i in range(n_iter):
        counter += 1
        if counter == k:
            out.append((i, k))
            counter = 0
            k = max(k - size_decr, k_min)
    return out


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dense_rows(x: torch.Tensor) -> bool:
    """True if ``x`` is [B, ...] with each sample one contiguous run of E elements."""
    if x.dim() < 2:
        return False
    if x.is_contiguous():
        return True
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return True
    if x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d):
        return True
    return False


LOSS_KIND = {'ce': 0, 'dlr': 1}


def _loss_pred(logits: torch.Tensor, y_hard, y_soft, loss_out, pred_out, want_dlogits: bool, kind: int = 0,
               y_target=None):
    """K2: per-sample CE, prediction and d(sum CE)/dlogits (``:113, 181-185, 194-197``).

    Kept as a module-level function so tests can substitute recorded losses.
    """
    lib = _lib.load()
    if logits.dim() != 2:
        raise _lib.ApgdHipError(f"model output must be [B, n_cls], got {tuple(logits.shape)}")
    lg = logits.detach()
    if lg.stride(1) != 1:
        lg = lg.contiguous()
    dl = torch.empty_like(lg) if want_dlogits else None
    if dl is not None and dl.stride() != lg.stride():
        dl = torch.empty_strided(lg.shape, lg.stride(), dtype=lg.dtype, device=lg.device)
    if kind == 2:                                    # dlr_loss_targeted (:106-111)
        _lib.check(lib.apgd_loss_pred_targeted(lg.data_ptr(), _lib.dtype_code(lg.dtype),
                                               lg.stride(0) if lg.shape[0] > 1 else lg.shape[1], y_hard.data_ptr(),
                                               y_target.data_ptr(), loss_out.data_ptr(), pred_out.data_ptr(), _lib.ptr(dl),
                                               lg.shape[0], lg.shape[1], _stream_ptr()), "apgd_loss_pred_targeted")
        return dl
    _lib.check(lib.apgd_loss_pred(lg.data_ptr(), _lib.dtype_code(lg.dtype), lg.stride(0) if lg.shape[0] > 1 else lg.shape[1],
                                  _lib.ptr(y_hard), _lib.ptr(y_soft), kind,
                                  loss_out.data_ptr(), pred_out.data_ptr(), _lib.ptr(dl),
                                  lg.shape[0], lg.shape[1], _stream_ptr()), "apgd_loss_pred")
    return dl


class ApgdWorkspace:
    """Per-call device state of the attack: three rotating iterates,
    best points, per-sample scalars.  All buffers share ``x``'s memory format."""

    def __init__(self, x: torch.Tensor, n_iter: int, n_rot: int):
        B = x.shape[0]
        dev = x.device
        self.rot = [torch.empty_like(x) for _ in range(n_rot)]
        self.x_best = torch.empty_like(x)
        self.x_best_adv = torch.empty_like(x)
        f32 = dict(device=dev, dtype=torch.float32)
        self.loss = torch.empty(B, **f32)
        self.loss_best = torch.empty(B, **f32)
        self.loss_best_last = torch.empty(B, **f32)
        self.reduced_last = torch.ones(B, **f32)                    # :201
        self.loss_steps = torch.zeros(max(n_iter, 1), B, **f32)     # :144
        self.pred = torch.empty(B, device=dev, dtype=torch.uint8)
        self.acc = torch.empty(B, device=dev, dtype=torch.uint8)
        # flag bytes of every iteration (apgd_state_update): row i drives the row moves of iteration i, which the Linf step of
        # iteration i + 1 performs (apgd_linf_step_track_f32); bench.py reads them back for the algorithmic-byte count
        self.flags = torch.zeros(max(n_iter, 1), B, device=dev, dtype=torch.uint8)


def _model_fwd_bwd_inner(model, x_in: torch.Tensor, y_hard, y_soft, ws: ApgdWorkspace, loss_out, pred_out,
                   need_grad: bool, kind: int = 0, y_target=None, sign_ok: bool = False, sign_blocked: bool = False):
    """One model call of the attack: forward, K2, and (optionally) the input gradient.

    ``autopgd_train_clean.py:174-192`` (first call) and ``:266-287`` (in-loop calls; the last
    one skips the backward, ``:281-283``).
    """
    if need_grad:
        x_in.requires_grad_(True)
        with torch.enable_grad(), ops.attack_forward():          # the backward below asks for the input gradient only
            logits = model(x_in)
        dl = _loss_pred(logits, y_hard, y_soft, loss_out, pred_out, True, kind, *(() if y_target is None else (y_target,)))
        if dl.shape != logits.shape or dl.dtype != logits.dtype:
            raise _lib.ApgdHipError("dlogits/logits mismatch")
        # the sink is opened for Linf only: the L2 step needs the gradient's values
        use_sink = sign_ok and USE_SIGN_SINK and not _sink_refused(model)
        sink = ops.grad_sign_sink(x_in, blocked=sign_blocked) if use_sink else contextlib.nullcontext()
        with ops.input_grad_only(), sink:
            grad = torch.autograd.grad([logits], [x_in], grad_outputs=[dl.view_as(logits)])[0].detach()
        x_in.requires_grad_(False)
        if getattr(sink, "signs", None) is not None:
            # The sink handed autograd a stride-0 zero in place of the stem's input gradient.  If that is ALL autograd returns,
            # the stem convolution was the only consumer of the iterate and its signs are the gradient's signs.  Anything else
            # (a dense tensor: autograd summed a second path from the raw input - an input skip, an ensemble sharing x) means
            # the signs are not the whole story: this model is taken off the sink and the call repeated on the fp32 path.
            if all(st == 0 for st in grad.stride()) or grad.numel() == 0:
                return sink.signs                            # int8 sign(grad), x_in's shape and (contiguous) layout
            import warnings
            warnings.warn("the attack iterate feeds more than the ConvStem's first convolution: gradient-sign sink disabled for "
                          "this model (fp32 input gradient through autograd)")
            try:
                _SINK_REFUSED.add(model)
            except TypeError:
                pass
            return _model_fwd_bwd_inner(model, x_in, y_hard, y_soft, ws, loss_out, pred_out, need_grad, kind, y_target, False, False)
        if grad.stride() != x_in.stride():
            g2 = torch.empty_like(x_in, dtype=grad.dtype)
            g2.copy_(grad)
            grad = g2
        return grad
    with torch.no_grad():
        logits = model(x_in)
    _loss_pred(logits, y_hard, y_soft, loss_out, pred_out, False, kind, *(() if y_target is None else (y_target,)))
    return None


_TWO_STREAM = weakref.WeakKeyDictionary()


def two_stream_model(model) -> bool:
    """Does this model ask for the two-stream form of a captured attack (``architecture.ConvNeXt.apgd_two_streams``: the narrow
    pyramids)?  When such a model's attack runs as overlapping batch chunks - and in every call ``graphed.run`` makes for it, warm-up
    or replay, so that they agree bit for bit - its GEMMs all run on cnx_gemm_nt (``ops.attack_pass``): overlapping chunks must not
    contain library GEMMs.  The plain eager attack (``graph=False``, the default) keeps the library's GEMMs where they are faster."""
    try:
        return _TWO_STREAM[model]
    except (KeyError, TypeError):
        pass
    v = bool(isinstance(model, torch.nn.Module) and any(getattr(m, "apgd_two_streams", False) for m in model.modules()))
    try:
        _TWO_STREAM[model] = v
    except TypeError:
        pass
    return v


def _model_fwd_bwd(model, *args, attack_gemm=False, **kw):
    if attack_gemm:
        with ops.attack_pass():
            return _model_fwd_bwd_inner(model, *args, **kw)
    return _model_fwd_bwd_inner(model, *args, **kw)


_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (device.index, n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


def _model_fwd_bwd_split(model, x_in, y_hard, y_soft, ws, loss_out, pred_out, need_grad, kind=0, y_target=None, sign_ok=False,
                         sign_blocked=False, splits=1, attack_gemm=False):
    """``_model_fwd_bwd`` with the batch cut into ``splits`` chunks that run on their own HIP streams (fork / join around the call).

    The attack treats samples independently and the model is in eval mode, so the chunks are independent problems with
    identical per-sample results; what changes is how the GPU is filled: at the late stages of ConvNeXt-T a kernel has 392
    (C = 384) or 98 (C = 768) workgroups for 256 CUs, and two streams let the tail of one chunk's kernel overlap the next
    kernel of the other (round 2 measured 185 -> 152 us per C = 384 forward this way and dropped it because the HOST could not
    feed two streams; under graph replay - graphed.py, the only caller with splits > 1 - there is no host in the loop)."""
    B = x_in.shape[0]
    if splits <= 1 or B < 2 * splits:
        return _model_fwd_bwd(model, x_in, y_hard, y_soft, ws, loss_out, pred_out, need_grad, kind, y_target, sign_ok, sign_blocked,
                              attack_gemm=attack_gemm)
    if not attack_gemm:
        raise _lib.ApgdHipError("overlapping batch chunks need the attack's GEMMs on cnx_gemm_nt (attack_gemm=True)")
    main = torch.cuda.current_stream()
    streams = _side_streams(x_in.device, splits)
    cuts = [B * i // splits for i in range(splits + 1)]
    full = None
    if need_grad:                                            # allocated on the calling stream, before the fork
        sink_expected = sign_ok and USE_SIGN_SINK and not _sink_refused(model)
        full = torch.empty(x_in.shape, device=x_in.device, dtype=torch.int8 if sink_expected else torch.float32)
        if x_in.dim() == 4 and not x_in.is_contiguous():
            full = torch.empty_like(x_in, dtype=full.dtype)
    parts = []
    ops._CHUNKED = True                                      # derived weight copies made by one chunk are awaited by the other (ops._cached)
    try:
        for h, st in enumerate(streams):
            a, b = cuts[h], cuts[h + 1]
            st.wait_stream(main)
            with torch.cuda.stream(st):                      # (chunks overlap: two_stream_model keeps library GEMMs out of them)
                g = _model_fwd_bwd(model, x_in[a:b], None if y_hard is None else y_hard[a:b], None if y_soft is None else y_soft[a:b],
                                   ws, loss_out[a:b], pred_out[a:b], need_grad, kind, None if y_target is None else y_target[a:b],
                                   sign_ok, sign_blocked, attack_gemm=True)
                if g is not None and g.dtype == full.dtype:
                    full[a:b].copy_(g)
                parts.append(g)
    finally:
        ops._CHUNKED = False
    for st in streams:
        main.wait_stream(st)
    if not need_grad:
        return None
    if any(g.dtype != full.dtype for g in parts):            # a chunk left the sign sink (second consumer detected): fp32 for all
        if all(g.dtype == parts[0].dtype for g in parts):
            full = torch.cat(parts, 0)
        else:
            full = torch.cat([g.float() if g.dtype == torch.int8 and not getattr(g, "apgd_blocked", False) else ops.signs_to_linear(g).float()
                              for g in parts], 0)
    elif full.dtype == torch.int8 and getattr(parts[0], "apgd_blocked", False):
        full.apgd_blocked = True
    return full


def apgd_train(model, x, y, norm, eps, n_iter=10, use_rs=False, loss='ce',
               verbose=False, mixup=None, is_train=True, graph=None):
    """Drop-in for the reference's ``apgd_train`` (``autopgd_train_clean.py:123-124``).

    Returns ``(x_best, acc, loss_best, x_best_adv)`` (``:371``): fresh, detached tensors with
    ``x``'s shape and memory format; ``acc`` is bool ``[B]``, ``loss_best`` fp32 ``[B]``.
    ``y`` is int64 ``[B]``, or fp32 ``[B, n_cls]`` probabilities iff ``mixup is not None``
    (``:194-197``).  Supported: ``norm in {'Linf', 'L2', 'L1'}`` (L1: ``loss='ce'``, see ``apgd_l1.py``), ``loss in {'ce', 'dlr'}``.

    ``graph`` (an addition of this path; default: the ``APGD_GRAPH`` environment variable, off): replay the attack - static
    shapes, no host synchronisation - from hipGraphs captured on the third call with a given (model, shapes, arguments), so
    that the host enqueues a handful of launches per attack instead of several hundred (``graphed.py``).  Same kernels, same
    results; the model's parameters are read at replay time.
    """
    assert not model.training                                           # :125
    if use_rs:
        # the reference executes `raise NotImplemented` here (:137); any exception type would do
        raise NotImplementedError("use_rs=True is not supported (autopgd_train_clean.py:137)")
    if loss not in criterion_names:
        raise KeyError(loss)                                            # criterion_dict[loss] (:149)
    if loss not in LOSS_KIND:
        # 'softloss' returns a scalar and 'dlr-targeted' takes 3 arguments: neither can be driven by the
        # reference's own loop (criterion_indiv(logits, y) then .sum(), :181-182) either
        raise NotImplementedError(f"loss={loss!r}: the HIP path implements 'ce' and 'dlr'")
    kind = LOSS_KIND[loss]
    if kind == 1 and mixup is not None:
        raise NotImplementedError("loss='dlr' needs hard labels (dlr_loss indexes x[arange, y], :103)")
    if norm == 'L1':                                                   # :160-167, 239-250, 351-362 ("next" row: apgd_l1.py)
        if kind != 0:
            raise NotImplementedError("norm='L1' is built for loss='ce'")
        from . import apgd_l1
        return apgd_l1.apgd_l1(model, x, y, eps, n_iter, mixup is not None, is_train, verbose, _model_fwd_bwd, ApgdWorkspace,
                               _stream_ptr)
    if norm not in ('Linf', 'L2'):
        raise NotImplementedError(f"norm={norm!r}: the HIP path covers Linf, L2 and L1 (L0 is broken in the reference itself, :257)")
    if graph is None:
        graph = GRAPH_DEFAULT
    if graph and not verbose:
        from . import graphed
        return graphed.run(model, x, y, norm, eps, n_iter, kind, mixup is not None)
    return _apgd_core(model, x, y, norm, eps, n_iter, kind, soft=mixup is not None, verbose=verbose)


def _apgd_core(model, x, y, norm, eps, n_iter, kind, soft=False, verbose=False, y_target=None, x_init=None, rec=None, splits=1,
               attack_gemm=None):
    """The device loop shared by ``apgd_train`` and the evaluation attacks (``aa_eval.apgd_attack``).

    ``x_init`` (optional) replaces the clean image as the start point (AutoAttack's random start); the ball stays
    centred on ``x``.  ``kind`` 2 = targeted DLR with ``y_target``.  ``rec`` (``graphed._Recorder``) is set while the loop is
    being captured into hipGraph segments: the update-kernel launches are handed to it as closures and stay outside the graphs.
    ``splits`` > 1: the model calls run as that many batch chunks on their own streams; ``attack_gemm`` (default: ``splits > 1``):
    every GEMM of the model calls on cnx_gemm_nt (``ops.attack_pass``) - ``graphed.run`` sets it for all its calls of a two-stream model.
    """
    if attack_gemm is None:
        attack_gemm = splits > 1
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise _lib.ApgdHipError("apgd_train needs a device (MI355X) tensor; there is no CPU fallback")
    if x.dtype != torch.float32:
        raise _lib.ApgdHipError(f"attack state is fp32 (got {x.dtype})")
    lib = _lib.load()
    n_iter = int(n_iter)
    if n_iter < 0:
        raise ValueError("n_iter must be >= 0")
    # n_iter == 0: `range(0)` in the reference (:209) - one forward/backward, then the clamped clean point with its
    # acc / loss is returned; the loop below simply does not run

    x = x.detach()
    if not _dense_rows(x):
        x = x.contiguous()
    B = x.shape[0]
    E = x[0].numel() if B > 0 else 0
    stream = _stream_ptr()
    if soft:
        y_soft, y_hard = y.detach().to(torch.float32).contiguous(), None
    else:
        y_hard, y_soft = y.detach().to(torch.int64).contiguous(), None

    ws = ApgdWorkspace(x, n_iter, n_rot=3 if n_iter > 1 else 2)
    cur = ws.rot[0]
    start = x
    if x_init is not None:
        start = x_init.detach().to(torch.float32)
        if start.shape != x.shape or start.stride() != x.stride():
            start = torch.empty_like(x).copy_(x_init)
    if y_target is not None:
        y_target = y_target.detach().to(torch.int64).contiguous()
    # Linf: the update kernel also makes the row moves of the iteration before it (and, at i = 0, the prologue's clones), see
    # apgd_linf_step_track_f32; the blocked sign order and L2 keep the separate tracking pass
    fused = norm == 'Linf' and n_iter > 0 and FUSED_TRACKING
    _lib.check(lib.apgd_init_f32(start.data_ptr(), cur.data_ptr(), None if fused else ws.x_best.data_ptr(),
                                 None if fused else ws.x_best_adv.data_ptr(), x.numel(), stream), "apgd_init_f32")  # :135, 141-143
    alpha = 2.0                                                              # :159
    step_size = torch.full((B,), alpha * eps, device=x.device, dtype=torch.float32)   # :169-170
    sched = dict(checkpoint_schedule(n_iter))
    l2_ws = None
    if norm == 'L2':
        l2_ws = torch.empty(3 * B * lib.apgd_l2_parts(), device=x.device, dtype=torch.float32)

    # first forward/backward: acc, loss_best are written directly by the loss kernel (:194-200)
    sign_ok = norm == 'Linf'
    grad = _model_fwd_bwd_split(model, cur, y_hard, y_soft, ws, ws.loss_best, ws.acc, True, kind, y_target, sign_ok, sign_ok, splits,
                                attack_gemm)
    if fused and K1_SHAPE is not None:
        raise _lib.ApgdHipError("apgd.K1_SHAPE (launch-shape sweep of the plain update kernel) has no effect with FUSED_TRACKING: set "
                                "apgd.FUSED_TRACKING = False for the sweep")
    grad_best = torch.empty_like(grad)                                       # :189
    if fused and getattr(grad, "apgd_blocked", False):
        # (blocked signs, APGD_SIGN_BLOCKED=1: an experiment of round 3, slower) the prologue's clones after all
        fused = False
        ws.x_best.copy_(cur)
        ws.x_best_adv.copy_(cur)
    if not fused:
        grad_best.copy_(grad)
    ws.loss_best_last.copy_(ws.loss_best)                                    # :200
    old = cur                                                                # :205 (x_adv_old == x_adv)
    free = [ws.rot[1]] + ([ws.rot[2]] if n_iter > 1 else [])

    for i in range(n_iter):                                                  # :209
        a = 0.75 if i > 0 else 1.0                                           # :218
        out = free.pop()
        if norm == 'Linf':
            fl_prev = ws.flags[i - 1] if (fused and i > 0) else None         # the flag bytes iteration i - 1 wrote
            def k1(i=i, a=a, cur=cur, old=old, grad=grad, out=out, fl_prev=fl_prev):
                # (the stream is looked up when the closure RUNS: under graph replay that is not the capture stream)
                if PROFILE_EVENTS is not None:
                    fl = fl_prev.clone() if fl_prev is not None else None            # what this launch will read (outside the bracket)
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record()
                g_code = _lib.I8_BLK if getattr(grad, "apgd_blocked", False) else _lib.dtype_code(grad.dtype)
                if fused:
                    _lib.check(lib.apgd_linf_step_track_f32(x.data_ptr(), cur.data_ptr(), old.data_ptr(), grad.data_ptr(), g_code,
                                                            step_size.data_ptr(), out.data_ptr(),
                                                            _lib.ptr(fl_prev), ws.x_best.data_ptr(),
                                                            grad_best.data_ptr(), ws.x_best_adv.data_ptr(), B, E, eps, a,
                                                            _stream_ptr()), "apgd_linf_step_track_f32")   # :214-226 + :304, 322-323, 345-346
                elif K1_SHAPE is not None and g_code != _lib.I8_BLK:
                    _lib.check(lib.apgd_linf_step_f32_ex(x.data_ptr(), cur.data_ptr(), old.data_ptr(), grad.data_ptr(), g_code,
                                                         step_size.data_ptr(), out.data_ptr(), None, B, E, eps, a, int(K1_SHAPE[0]),
                                                         int(K1_SHAPE[1]), int(K1_SHAPE[2]), _stream_ptr()), "apgd_linf_step_f32_ex")
                else:
                    _lib.check(lib.apgd_linf_step_f32(x.data_ptr(), cur.data_ptr(), old.data_ptr(), grad.data_ptr(), g_code,
                                                      step_size.data_ptr(), out.data_ptr(), None, B, E, eps, a, _stream_ptr()),
                               "apgd_linf_step_f32")                         # :214-226
                if PROFILE_EVENTS is not None:
                    ev1 = torch.cuda.Event(enable_timing=True)
                    ev1.record()
                    PROFILE_EVENTS.append(("apgd_linf_step_track_f32" if fused else "apgd_linf_step_f32", i, ev0, ev1,
                                           grad.element_size(), fl))
            if rec is not None:
                rec.eager(k1)
            else:
                k1()
        else:
            _lib.check(lib.apgd_l2_step_f32(x.data_ptr(), cur.data_ptr(), old.data_ptr(), grad.data_ptr(),
                                            step_size.data_ptr(), out.data_ptr(), l2_ws.data_ptr(), B, E, eps, a,
                                            stream), "apgd_l2_step_f32")     # :229-237
        if old is not cur:
            free.append(old)
        old, cur = cur, out                                                  # :215, 260 (buffer rotation)

        last = i == n_iter - 1
        g_new = _model_fwd_bwd_split(model, cur, y_hard, y_soft, ws, ws.loss, ws.pred, not last, kind, y_target, sign_ok, sign_ok,
                                     splits, attack_gemm)                    # :266-287
        if g_new is not None:
            if g_new.dtype != grad_best.dtype:               # a model that switches gradient form mid-attack
                g_new = torch.sign(g_new).to(grad_best.dtype) if grad_best.dtype == torch.int8 else g_new.to(grad_best.dtype)
                if getattr(grad, "apgd_blocked", False):     # grad_best rows are in blocked order: so must be the new gradient
                    g_new = ops.signs_to_blocked(g_new)
            grad = g_new

        do_check = i in sched                                                # :329
        k = sched.get(i, 1)
        _lib.check(lib.apgd_state_update(ws.loss.data_ptr(), ws.pred.data_ptr(), ws.acc.data_ptr(),
                                         ws.loss_best.data_ptr(), ws.loss_best_last.data_ptr(),
                                         ws.reduced_last.data_ptr(), step_size.data_ptr(), ws.loss_steps.data_ptr(),
                                         ws.flags[i].data_ptr(), B, n_iter, i, int(do_check), k, float(k * 0.75), stream),
                   "apgd_state_update")                                      # :296, 319-343
        if last or not fused:                                # (fused: iteration i's row moves ride on the step of iteration i + 1)
            _lib.check(lib.apgd_track_rows(ws.flags[i].data_ptr(), cur.data_ptr(), grad.data_ptr(), ws.x_best.data_ptr(),
                                           grad_best.data_ptr(), ws.x_best_adv.data_ptr(), grad.element_size(), B, E,
                                           int(last), stream), "apgd_track_rows")  # :304, 322-323, 345-346
        if verbose:                                                          # :306-311 (host sync, debug only)
            print('iteration: {} - best loss: {:.6f} curr loss {:.6f} - robust accuracy: {:.2%} - step size: {:.5f}'.format(
                i, ws.loss_best.sum().item(), ws.loss.mean().item(), ws.acc.float().mean().item(),
                step_size.mean().item()))

    return ws.x_best, ws.acc.view(torch.bool), ws.loss_best, ws.x_best_adv   # :371
