"""Outer adversarial-training step (``/root/reference/main.py:961-997``) and its DDP wiring
(``main.py:351-359, 889-890``): per-iteration LR, ``zero_grad(set_to_none)``, autocast forward
through ``WrappedModel`` (attack inside), loss, backward (RCCL gradient all-reduce fired by DDP's
hooks — the only collective of the path), AdamW step, EMA update.

Differences from the reference, all MI355X-motivated (SURVEY.md §7 "hard parts"):
  * bf16 autocast, no GradScaler (the reference uses fp16 + GradScaler, ``main.py:797, 992-994``);
  * the EMA copy lives on the device and is updated with one multi-tensor lerp instead of a
    per-step device->host copy of every parameter (``main.py:885, 996-997``).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import graphed, ops
from .config import AdvConfig, wrap_model_for_at


def get_cosine_lr(epoch, lr, epochs, lr_peak_epoch):
    """Linear warm-up to ``lr`` then cosine to 5e-6 (``main.py:227-243``)."""
    if epoch <= lr_peak_epoch:
        return float(np.interp([epoch], [0, lr_peak_epoch], [1e-4 * lr, lr])[0])
    lr_min = 5e-6
    return lr_min + .5 * (lr - lr_min) * (1 + math.cos(math.pi * (epoch - lr_peak_epoch) / (epochs - lr_peak_epoch)))


def iteration_lrs(epoch, iters, **kw):
    """Per-iteration LR table of one epoch (``main.py:956-958``)."""
    return np.interp(np.arange(iters), [0, iters], [get_cosine_lr(epoch, **kw), get_cosine_lr(epoch + 1, **kw)])


def create_optimizer(model: nn.Module, arch: str, weight_decay: float = 0.05, capturable: bool = False):
    """AdamW(betas=(0.9, 0.95)) with the reference's two parameter groups (``main.py:395-459``):
    for convnext/resnet archs the names containing ``bn`` or ``.bias`` are not decayed (LayerNorm
    weights and gammas ARE); otherwise 1-D parameters and biases are not decayed.

    ``capturable``: step counters and the learning rate are device tensors, so the update can be part of a hipGraph
    (``ATTrainStep(graph_train=True)``); the arithmetic is the same."""
    named = [(k, v) for k, v in model.named_parameters() if v.requires_grad]
    if 'convnext' in arch or 'resnet' in arch:
        excluded = ['bn', '.bias']
        no_decay = [v for k, v in named if any(c in k for c in excluded)]
        decay = [v for k, v in named if not any(c in k for c in excluded)]
    else:
        no_decay = [v for k, v in named if v.ndim <= 1 or k.endswith('.bias')]
        decay = [v for k, v in named if not (v.ndim <= 1 or k.endswith('.bias'))]
    groups = [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]
    if capturable:
        dev = decay[0].device
        return torch.optim.AdamW(groups, lr=torch.tensor(1e-3, dtype=torch.float32, device=dev), betas=(0.9, 0.95), fused=True,
                                 capturable=True)
    return torch.optim.AdamW(groups, betas=(0.9, 0.95), fused=decay[0].is_cuda)


class DeviceEma:
    """``ModelEmaV2(decay)`` semantics (``ema = decay*ema + (1-decay)*model`` over the whole state dict,
    ``main.py:882-887, 996-997``) kept on the device and updated by one multi-tensor lerp.

    The copy keeps the model's state-dict keys (``base_model.*`` under ``WrappedModel``), which is what the
    reference saves as ``weights_ema_{epoch}.pt`` / ``state_dict_ema`` via ``get_state_dict(model_ema)``
    (``main.py:739-747``).  Non-floating entries (none in the LayerNorm models of this path, but e.g.
    ``num_batches_tracked`` elsewhere) are copied verbatim at every update."""

    def __init__(self, model: nn.Module, decay: float = 0.9999):
        self.decay = decay
        sd = model.state_dict()
        self.keys = list(sd.keys())
        self._model_sd = sd                                   # live views of the parameters / buffers
        self._ema_sd = {k: v.detach().clone() for k, v in sd.items()}
        self._fkeys = [k for k in self.keys if sd[k].is_floating_point()]
        self._ikeys = [k for k in self.keys if not sd[k].is_floating_point()]
        self.src = [sd[k] for k in self._fkeys]
        self.ema = [self._ema_sd[k] for k in self._fkeys]

    @torch.no_grad()
    def update(self):
        torch._foreach_lerp_(self.ema, [s.detach() for s in self.src], 1.0 - self.decay)
        for k in self._ikeys:
            self._ema_sd[k].copy_(self._model_sd[k])

    def state_dict(self):
        """``{key: ema tensor}`` with the wrapped model's own keys (timm ``get_state_dict(model_ema)``)."""
        return {k: self._ema_sd[k] for k in self.keys}

    @torch.no_grad()
    def load_state_dict(self, sd, strict: bool = True):
        missing = [k for k in self.keys if k not in sd]
        extra = [k for k in sd if k not in self._ema_sd]
        if strict and (missing or extra):
            raise KeyError(f"EMA state dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in self.keys:
            if k in sd:
                self._ema_sd[k].copy_(sd[k])


def setup_distributed():
    """One process per GPU, RCCL via the "nccl" backend (``main.py:351-356``); reads the torchrun env."""
    import torch.distributed as dist
    # dmabuf IPC (the only mode the host driver of this pool supports): must be in the environment before the HIP runtime
    # starts in this process, or RCCL's buffer exchange fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # RCCL ("nccl") on GPUs; APGD_DIST_BACKEND=gloo lets the same wiring be exercised with several ranks on ONE GPU
        # (RCCL refuses two ranks per device), which is how the 1-GPU test box covers DDP + the hand-written backward
        backend = os.environ.get("APGD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class _TrainPassGraph:
    """The training pass of one step - forward of the adversarial batch, loss, backward, AdamW, EMA (``main.py:985-997``) -
    captured once as a hipGraph and replayed: ~350 kernel launches through Python autograd functions become one graph
    launch.  Same capture rules as ``graphed._Program``: parameters (and their ``.grad``, allocated in the graph's pool by
    the captured backward) are read and written in place, derived weight copies are rebuilt inside the graph."""

    def __init__(self, step: "ATTrainStep", x, target, x_is_static: bool = False):
        # x_is_static: x is the static output of the attack's own graph (graphed.borrow_outputs) - the same tensor at every
        # step, read in place; anything else is copied into a buffer of this graph
        self.x = x if x_is_static else torch.empty_like(x)
        self.t = torch.empty_like(target)
        if not x_is_static:
            self.x.copy_(x)
        self.t.copy_(target)
        self.derived = {}
        self.graph = torch.cuda.CUDAGraph()
        step.optimizer.zero_grad(set_to_none=True)             # the captured backward allocates every .grad in the graph's pool
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        prev = ops._CAPTURE_CACHE
        ops._CAPTURE_CACHE = self.derived
        try:
            with torch.cuda.stream(side):
                self.graph.capture_begin(capture_error_mode=graphed.capture_mode())
                try:
                    self.loss = step._train_pass(self.x, self.t)
                except BaseException:
                    try:
                        self.graph.capture_end()
                    except Exception:                          # noqa: BLE001 - the capture is already invalid
                        pass
                    raise
                self.graph.capture_end()
        finally:
            ops._CAPTURE_CACHE = prev
        torch.cuda.current_stream().wait_stream(side)

    def __call__(self, x, target):
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.t.copy_(target)
        self.graph.replay()
        return self.loss.clone()


TRAIN_GRAPH_WARMUP = 3            # eager steps per batch shape before its training pass is captured (the attack's graphs: step 3)


class ATTrainStep:
    """Builds ``DDP(WrappedModel(model, apgd))`` + optimizer (+EMA) and runs single steps.

    ``graph_train`` (default: on when ``adv.graph`` is set, on one GPU): after ``TRAIN_GRAPH_WARMUP`` eager steps the
    training pass is replayed from a hipGraph (``_TrainPassGraph``); it needs static batch shapes - a batch of another shape
    runs eagerly.  Under DDP the pass stays eager (the gradient all-reduce is fired from autograd hooks)."""

    def __init__(self, model: nn.Module, arch: str, adv: AdvConfig, device, lr: float = 1e-3,
                 weight_decay: float = 0.05, distributed: bool = False, channels_last: bool = True,
                 amp_dtype: Optional[torch.dtype] = torch.bfloat16, ema: bool = True, mixup=None,
                 soft_targets: bool = False, perturb=None, gemm_table: bool = False, ema_decay: float = 0.9999,
                 graph_train: Optional[bool] = None):
        self.device = torch.device(device)
        if graph_train is None:
            graph_train = bool(getattr(adv, "graph", 0)) and os.environ.get("APGD_GRAPH_TRAIN", "1") != "0"
        self.graph_train = bool(graph_train) and not distributed and self.device.type == 'cuda'
        self._tg = {}                                                      # (shapes, dtypes) -> _TrainPassGraph | None (failed)
        self._tg_seen = {}                                                 # (shapes, dtypes) -> eager steps so far
        if gemm_table and self.device.type == 'cuda':
            # opt-in: points PyTorch's process-wide TunableOp at the shipped, read-only hipBLASLt solution table
            ops.load_gemm_table()
        if channels_last:
            model = model.to(memory_format=torch.channels_last)            # main.py:815-817
        if perturb is not None:                                            # any callable(model, x, y), as main.py:844
            from .wrapped_model import WrappedModel
            wrapped = WrappedModel(model, perturb).to(self.device)
        else:
            wrapped = wrap_model_for_at(model, adv, mixup=mixup).to(self.device)  # main.py:831-844, 881
        self.ema = DeviceEma(wrapped, ema_decay) if ema else None                     # before DDP (main.py:882-887)
        self.perturb = adv.attack != 'none' or perturb is not None
        self.inner = wrapped
        if distributed:
            ids = [self.device.index] if self.device.type == 'cuda' else None
            # main.py:889-890.  The models of this path carry constant buffers only (ImageNormalizer mean / std), so the
            # per-forward buffer broadcast is dropped; gradients live in the all-reduce buckets (no copy in / out per step).
            wrapped = nn.parallel.DistributedDataParallel(wrapped, device_ids=ids, broadcast_buffers=False,
                                                          gradient_as_bucket_view=True)
        self.model = wrapped
        self.optimizer = create_optimizer(self.inner, arch, weight_decay, capturable=self.graph_train)
        self.loss = (lambda o, t: torch.sum(-t * torch.log_softmax(o.float(), dim=-1), dim=-1).mean()) \
            if soft_targets else nn.CrossEntropyLoss()                     # SoftTargetCrossEntropy / CE (main.py:461-466)
        self.amp_dtype = amp_dtype
        self.lr = lr
        self.mixup_fn = mixup if callable(mixup) else None                # timm-style Mixup object (main.py:599-607)
        if self.perturb:
            self.inner.set_perturb(True)                                   # main.py:950-954
        self.model.train()

    def _set_lr(self, lr):
        for g in self.optimizer.param_groups:                              # main.py:973-974
            if isinstance(g['lr'], torch.Tensor):
                g['lr'].fill_(lr)                                          # capturable optimizer: the graph reads this tensor
            else:
                g['lr'] = lr

    def _train_pass(self, x_adv, target):
        """forward - loss - backward - AdamW - EMA on an already perturbed batch (what ``_TrainPassGraph`` captures)."""
        with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            output = self.inner.base_model(x_adv)                          # main.py:293 (WrappedModel.forward after the attack)
            loss = self.loss(output, target)
        loss.backward()
        self.optimizer.step()
        if self.ema is not None:
            self.ema.update()
        return loss.detach()

    def _graph_step(self, images, target):
        """The step with the training pass replayed from a hipGraph; None if this batch has to run eagerly."""
        key = (tuple(images.shape), images.dtype, tuple(target.shape), target.dtype)
        if key in self._tg and self._tg[key] is None:
            return None
        if key not in self._tg and self._tg_seen.get(key, 0) < TRAIN_GRAPH_WARMUP:
            self._tg_seen[key] = self._tg_seen.get(key, 0) + 1            # libraries meet every shape outside a capture first
            return None
        base = self.inner.base_model
        replays0 = graphed.STATS["replays"]
        if self.perturb:                                                   # WrappedModel.forward, main.py:276-292 (under the
            base.eval()                                                    # step's autocast, as main.py:985 has it)
            with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None), \
                    graphed.borrow_outputs():
                z = self.inner.perturb(base, images, target)
            base.train()
            if isinstance(z, (tuple, list)):
                z = z[0]
        else:
            z = images
        prog = self._tg.get(key)
        if prog is None:
            try:
                # (a replayed attack under borrow_outputs hands out its graph's own static tensor: stable address)
                prog = self._tg[key] = _TrainPassGraph(self, z, target, x_is_static=graphed.STATS["replays"] > replays0)
            except Exception as e:                                         # noqa: BLE001 - any capture failure means "run eagerly"
                import warnings
                warnings.warn(f"training-pass graph capture failed ({type(e).__name__}: {e}); this batch shape runs eagerly")
                self._tg[key] = None
                torch.cuda.synchronize()
                self.optimizer.zero_grad(set_to_none=True)
                loss = self._train_pass(z, target)
                ops.invalidate_weight_cache()
                return loss
        loss = prog(z, target)
        ops.invalidate_weight_cache()                                      # eager consumers of packed / bf16 weight copies
        return loss

    def step(self, images, target, lr: Optional[float] = None):
        self._set_lr(self.lr if lr is None else lr)
        if self.mixup_fn is not None:
            images, target = self.mixup_fn(images, target)                 # main.py:965-966 (soft labels [B, n_cls])
        if self.graph_train and images.is_cuda:
            loss = self._graph_step(images, target)
            if loss is not None:
                return loss
        return self._eager_step(images, target)

    def _eager_step(self, images, target):
        self.optimizer.zero_grad(set_to_none=True)                         # main.py:984
        with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            output = self.model(images, target) if self.perturb else self.model(images)   # main.py:985-989
            loss = self.loss(output, target)                               # main.py:990
        loss.backward()                                                    # main.py:992 (DDP all-reduce inside)
        self.optimizer.step()                                              # main.py:993
        ops.invalidate_weight_cache()                                      # packed / bf16 weight copies follow the update
        if self.ema is not None:
            self.ema.update()                                              # main.py:996-997
        return loss.detach()
