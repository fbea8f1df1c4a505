"""hipGraph replay of the APGD attack (``apgd_train(..., graph=True)`` / ``APGD_GRAPH=1``).

Why.  One adversarial-training step of ConvNeXt-T enqueues ~900 kernels from Python (autograd functions + ctypes), 40-50 ms
of host time against 50-57 ms of GPU time: every kernel gain shrinks into that wall, and eight rank processes share one
host.  The attack is 3/5 of those launches, has static shapes and performs no host synchronisation
(``autopgd_train_clean.py:123-371`` on the device, ``apgd._apgd_core``), so it is captured once and replayed.

How.  The third call with a given (model, input signature, arguments) runs ``_apgd_core`` under stream capture.  The
Linf-update launches (K1) are NOT captured: ``_apgd_core`` hands them to the recorder as closures, which cuts the capture
into ``n_iter + 1`` graph segments with the K1 launches between them - ``bench.py`` brackets exactly those launches with HIP
events on the launch stream for its ``roofline`` object, which a node inside a graph does not allow.  A replay is then
``n_iter + 1`` graph launches and ``n_iter`` kernel launches.

What a replay reads.  The caller's ``x`` / ``y`` are copied into the static input buffers; model PARAMETERS are read in
place by the captured kernels, and every derived copy of a parameter (fragment-packed MLP weights, bf16 casts, transposed
filters: ``ops._cached``) is rebuilt INSIDE the graph from the live parameter - during capture ``ops`` uses a capture-local
cache - so an optimizer step between two replays needs no host-side refresh and nothing can go stale.  Results are returned
as fresh tensors (clones of the graph's static outputs), as ``apgd_train`` promises.

Capture failures (an operator that synchronises, a library that refuses capture) fall back to the eager loop with a warning:
same kernels, same results, only the host cost differs.
"""
from __future__ import annotations

import os
import warnings
import weakref

import torch
import torch.utils._python_dispatch

from . import apgd, ops

WARMUP_CALLS = 2                 # eager calls before the capture (library handles, kernel attributes, autotuned choices)
# batch chunks the model calls of a captured attack are cut into, each on its own stream (apgd._model_fwd_bwd_split): fills the CUs
# the late stages' small kernels leave idle; costs nothing on the host once captured.  APGD_ATTACK_STREAMS=1: one stream.
STREAMS = int(os.environ.get("APGD_ATTACK_STREAMS", "2"))


def _streams(model):
    """Chunks / streams of a captured attack: two for the models that ask for it (``apgd_two_streams``: the narrow ConvNeXt
    pyramids), else one.  With more than one stream the attack's GEMMs all run on cnx_gemm_nt (``ops.attack_pass``): two of the
    library's stream-K GEMMs in flight at once deadlocked the GPU (ViT-B shapes, round 3) - hence never with APGD_GEMM=lib /
    APGD_OPS=eager, never for a model that is not built from ``architecture``'s classes, and never outside bf16 autocast:
    cnx_gemm_nt takes bf16 operands only, so under fp32 / fp16 every GEMM of the chunks would be the library's.  What this static
    test cannot see (a layer whose shape fails the kernel's guards) the warm-up calls catch: ``_LibGemmWatch``."""
    if ops._GEMM_MODE == "lib" or ops.MODE == "eager" or STREAMS <= 1:
        return 1
    if not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return 1
    return STREAMS if apgd.two_stream_model(model) else 1


class _LibGemmWatch(torch.utils._python_dispatch.TorchDispatchMode):
    """Counts the library GEMMs (hipBLASLt / rocBLAS through ATen) a model call issues.  ``run`` puts the FIRST warm-up call of a
    would-be two-stream signature under it, on one stream: if anything in the attack's model calls still reaches the library
    although ``ops.attack_pass`` is on (a layer that fails cnx_gemm_nt's guards, a foreign submodule), the signature is
    captured and run on ONE stream - overlapping chunks never contain a library GEMM."""
    # GEMMs, and the library convolutions: MIOpen lowers 1x1, patchify and 2x2-stride-2 convolutions to rocBLAS / hipBLASLt GEMMs
    # (a foreign stem, a downsample layer that fails ops.downsample_supported) - two of those in flight are the same hazard
    NAMES = ("mm", "addmm", "bmm", "baddbmm", "linear", "matmul", "_scaled_mm", "addmv", "mv",
             "convolution", "convolution_backward", "_convolution", "cudnn_convolution", "miopen_convolution",
             "miopen_convolution_transpose", "miopen_depthwise_convolution", "convolution_overrideable",
             "convolution_backward_overrideable", "_convolution_double_backward")

    def __init__(self):
        super().__init__()
        self.count = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func)).split(".")[0]
        if name in self.NAMES and any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
            self.count += 1
        return func(*args, **(kwargs or {}))


STATS = {"captures": 0, "replays": 0, "eager": 0, "failed": 0, "evicted": 0, "lib_gemm_one_stream": 0}
# Captured programs, most recently used last.  Each holds a private graph pool (the attack state, the activations of its model
# passes, packed weight copies: multi-GB at batch 256), so their number is bounded: beyond MAX_PROGRAMS the least recently used
# capture is dropped (its signature starts over: warm-up calls, then a new capture), and a model's entries go when the model does.
MAX_PROGRAMS = int(os.environ.get("APGD_GRAPH_MAX_PROGRAMS", "4"))
_programs = {}
_finalizers = {}


def _drop_model(mid):
    for k in [k for k in _programs if k[0] == mid]:
        del _programs[k]
    _finalizers.pop(mid, None)


def _watch_model(model):
    mid = id(model)
    if mid not in _finalizers:
        try:
            _finalizers[mid] = weakref.finalize(model, _drop_model, mid)
        except TypeError:                                    # not weak-referenceable: the id check in run() still applies
            pass


def _evict():
    live = [k for k, e in _programs.items() if e["prog"] is not None]
    while len(live) > MAX_PROGRAMS:
        del _programs[live.pop(0)]                           # dicts keep insertion order; run() re-inserts on every use
        STATS["evicted"] += 1


def capture_mode():
    """hipStreamCaptureMode of our captures.  "global" (torch's default) makes an unsafe runtime call from ANY thread an error
    while a capture is open - including the event queries of the RCCL watchdog thread of a multi-GPU rank, which polls the
    gradient all-reduce of the previous step while this step's attack is being captured (the host runs ahead of the device).
    With a process group up, captures therefore use "thread_local" (only the capturing thread is policed), after a device
    synchronisation that leaves the watchdog nothing to poll."""
    import torch.distributed as dist
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


class _Recorder:
    """Cuts one pass of ``_apgd_core`` into graph segments separated by eager closures."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.steps = []                                      # torch.cuda.CUDAGraph | callable
        self._g = None
        self.mode = capture_mode()

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool, capture_error_mode=self.mode)

    def end(self):
        g, self._g = self._g, None
        g.capture_end()
        self.steps.append(g)

    def abort(self):
        """After an exception inside the capture: leave capture mode, whatever state the graph is in."""
        g, self._g = self._g, None
        if g is not None:
            try:
                g.capture_end()
            except Exception:                                # noqa: BLE001 - the capture is already invalid
                pass

    def eager(self, fn):
        self.end()
        self.steps.append(fn)
        ops._CAPTURE_SEG += 1                                # derived copies of earlier segments need no cross-stream wait (ops._cached)
        self.begin()


class _Program:
    def __init__(self, model, x, y, norm, eps, n_iter, kind, soft, splits, y_target=None, x_init=None):
        self.model_ref = weakref.ref(model)
        self.x = torch.empty_like(x)
        self.y = torch.empty_like(y)
        self.x.copy_(x)
        self.y.copy_(y)
        # evaluation attacks (aa_eval): target classes and the random start are inputs of the replay like x and y
        self.yt = None if y_target is None else y_target.clone()
        self.xi = None if x_init is None else torch.empty_like(x).copy_(x_init)
        self.derived = {}                                    # capture-local ops._cached entries (kept alive with the graphs)
        rec = _Recorder()
        if rec.mode != "global":
            torch.cuda.synchronize()                         # collectives of earlier steps are done: nothing for a watchdog to poll
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        prev = ops._CAPTURE_CACHE
        ops._CAPTURE_CACHE = self.derived
        try:
            with torch.cuda.stream(side):
                rec.begin()
                try:
                    self.out = apgd._apgd_core(model, self.x, self.y, norm, eps, n_iter, kind, soft=soft, rec=rec, splits=splits,
                                               attack_gemm=_attack_gemm(model), y_target=self.yt, x_init=self.xi)
                except BaseException:
                    rec.abort()
                    raise
                rec.end()
        finally:
            ops._CAPTURE_CACHE = prev
        torch.cuda.current_stream().wait_stream(side)
        self.steps = rec.steps
        self.n_graphs = sum(isinstance(s, torch.cuda.CUDAGraph) for s in self.steps)

    def __call__(self, x, y, y_target=None, x_init=None):
        self.x.copy_(x)
        self.y.copy_(y)
        if self.yt is not None:
            self.yt.copy_(y_target)
        if self.xi is not None:
            self.xi.copy_(x_init)
        for s in self.steps:
            if isinstance(s, torch.cuda.CUDAGraph):
                s.replay()
            else:
                s()
        if BORROW:                                           # the caller consumes the results before the next call (borrow_outputs)
            return self.out
        x_best, acc, loss_best, x_best_adv = self.out
        return x_best.clone(), acc.clone(), loss_best.clone(), x_best_adv.clone()


BORROW = False
# The program whose replay produced the most recent attack (None: that attack ran eagerly).  ATTrainStep reads it right behind its
# attack call: a training-pass graph captured behind a replay of program P may read P's derived weight copies (P.derived - rebuilt by
# every replay of P from the live parameters) instead of rebuilding its own, as long as P is what ran in front of it.
LAST = None


class borrow_outputs:
    """Inside this context a replayed attack returns its graph's static output tensors themselves instead of fresh copies
    (two 154 MB clones at the headline shapes): for a caller that consumes them before the next attack call - ``ATTrainStep``,
    whose training-pass graph reads ``x_best`` in place."""

    def __enter__(self):
        global BORROW
        self._prev = BORROW
        BORROW = True
        return self

    def __exit__(self, *exc):
        global BORROW
        BORROW = self._prev
        return False


def _signature(model, x, y, norm, eps, n_iter, kind, soft, y_target=None, x_init=None):
    ac = (torch.is_autocast_enabled(), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
    return (id(model), tuple(x.shape), tuple(x.stride()), x.dtype, x.device.index, tuple(y.shape), y.dtype, norm, float(eps),
            int(n_iter), int(kind), bool(soft), ac, ops.MODE, apgd.USE_SIGN_SINK, _streams(model), y_target is not None, x_init is not None)


def _attack_gemm(model):
    """Every call ``run`` makes for a would-be two-stream model - warm-up, capture, the eager fallback - keeps the attack's GEMMs
    on cnx_gemm_nt, so that they all agree bit for bit whatever the number of streams turns out to be."""
    return _streams(model) > 1


def reset():
    """Drop every captured program (tests; after replacing a model's parameters by new tensors; bench.py between configurations)."""
    _programs.clear()


def run(model, x, y, norm, eps, n_iter, kind, soft, y_target=None, x_init=None):
    """``_apgd_core`` with graph replay: eager for the first ``WARMUP_CALLS`` calls of a signature, captured on the next.
    ``y_target`` / ``x_init`` (the evaluation attacks of ``aa_eval``: target classes, random start) are replay inputs like x and y."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and isinstance(y, torch.Tensor) and y.is_cuda):
        return apgd._apgd_core(model, x, y, norm, eps, n_iter, kind, soft=soft, y_target=y_target, x_init=x_init)     # raises the usual errors
    global LAST
    LAST = None
    x = x.detach()
    if not apgd._dense_rows(x):
        x = x.contiguous()
    y = y.detach()
    ext = dict(y_target=y_target, x_init=x_init)
    key = _signature(model, x, y, norm, eps, n_iter, kind, soft, y_target, x_init)
    ent = _programs.pop(key, None)
    if ent is None:
        ent = {"calls": 0, "prog": None, "failed": False, "one_stream": False}
        _watch_model(model)
    _programs[key] = ent                                     # most recently used last
    prog = ent["prog"]
    if prog is not None and prog.model_ref() is not model:   # id() of a dead model handed to a new one
        ent.update(calls=0, prog=None, failed=False, one_stream=False)
        prog = None
    ag = _attack_gemm(model)
    if prog is None:
        splits = 1 if (ent["failed"] or ent["one_stream"]) else _streams(model)
        if ent["failed"] or ent["calls"] < WARMUP_CALLS:
            ent["calls"] += 1
            STATS["eager"] += 1
            if splits > 1 and ent["calls"] == 1:
                # first warm-up call of a two-stream signature: one stream, library GEMMs counted
                with _LibGemmWatch() as watch:
                    out = apgd._apgd_core(model, x, y, norm, eps, n_iter, kind, soft=soft, splits=1, attack_gemm=ag, **ext)
                if watch.count:
                    ent["one_stream"] = True
                    STATS["lib_gemm_one_stream"] += 1
                    warnings.warn(f"{watch.count} library GEMM calls inside the attack's model calls: this signature is captured on "
                                  "one stream (overlapping chunks must not contain library GEMMs)")
                return out
            # (same batch chunks as the capture will use: every kernel / library shape is initialised before it)
            return apgd._apgd_core(model, x, y, norm, eps, n_iter, kind, soft=soft, splits=splits, attack_gemm=ag, **ext)
        try:
            prog = ent["prog"] = _Program(model, x, y, norm, eps, n_iter, kind, soft, splits, y_target, x_init)
            STATS["captures"] += 1
            _evict()
        except Exception as e:                               # noqa: BLE001 - any capture failure means "run eagerly"
            ent["failed"] = True
            STATS["failed"] += 1
            warnings.warn(f"APGD graph capture failed ({type(e).__name__}: {e}); this signature runs eagerly")
            torch.cuda.synchronize()
            return apgd._apgd_core(model, x, y, norm, eps, n_iter, kind, soft=soft, attack_gemm=ag, **ext)
    STATS["replays"] += 1
    out = prog(x, y, y_target, x_init)
    LAST = prog
    return out
