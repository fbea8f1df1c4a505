"""``adv.attack=fgsm`` - the second value of the reference's attack selector (``/root/reference/main.py:836-842``), same
injection point as APGD (``WrappedModel.perturb``).  Drop-in for ``fgsm_train`` (``/root/reference/fgsm_train.py:72-100``):

    x_adv = x + (2 t - 1) eps noise_level, t ~ U[0,1)      (use_rs; clamped to [0, 1] unless skip_projection)
    g     = d sum CE(model(x_adv), y) / d x_adv
    x_adv = x_adv + alpha eps sign(g);   unless skip_projection:  x + clamp(x_adv - x, -eps, eps), clamped to [0, 1]

Device side: ``apgd_fgsm_start_f32`` / ``apgd_fgsm_step_f32`` (``csrc/apgd_kernels.hip``) around one forward / backward that
goes through the same loss kernel and int8 gradient-sign sink as the APGD path; no host synchronisation.  Bit-identical to the
reference given the same uniform draw and the same model outputs (``tests/golden/fgsm_*.npz``).
"""
from __future__ import annotations

import torch

from . import _lib
from . import apgd as _apgd


def fgsm_train(model, x, y, eps, loss='ce', alpha=1.25, use_rs=False, noise_level=1., skip_projection=False, _t=None):
    """Same signature and result as the reference's ``fgsm_train`` (one tensor, ``x``'s shape and memory format).
    ``y``: int64 ``[B]`` or float ``[B, n_cls]`` probabilities (mixup), as ``F.cross_entropy`` takes them.  ``loss`` must be
    ``'ce'`` (the reference's ``criterion_dict`` of that file has no other entry: ``KeyError``).  ``_t`` (tests only) replaces
    the uniform draw."""
    assert not model.training                                           # :74
    if loss != 'ce':
        raise KeyError(loss)                                            # criterion_dict[loss], fgsm_train.py:12, 86
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise _lib.ApgdHipError("fgsm_train needs a device (MI355X) tensor; there is no CPU fallback")
    if x.dtype != torch.float32:
        raise _lib.ApgdHipError(f"attack state is fp32 (got {x.dtype})")
    lib = _lib.load()
    x = x.detach()
    if not _apgd._dense_rows(x):
        x = x.contiguous()
    B = x.shape[0]
    stream = _apgd._stream_ptr()
    eps = float(eps)
    project = 0 if skip_projection else 1
    soft = y.dtype.is_floating_point
    if soft:
        y_soft, y_hard = y.detach().to(torch.float32).contiguous(), None
    else:
        y_hard, y_soft = y.detach().to(torch.int64).contiguous(), None
    if use_rs:
        t = torch.rand_like(x) if _t is None else _t.detach().to(torch.float32)     # :81 (the caller's generator, as in the reference)
        if t.stride() != x.stride():
            t = torch.empty_like(x).copy_(t)
        start = torch.empty_like(x)
        _lib.check(lib.apgd_fgsm_start_f32(x.data_ptr(), t.data_ptr(), start.data_ptr(), x.numel(), eps, float(noise_level), project,
                                           stream), "apgd_fgsm_start_f32")          # :82-84
    else:
        start = x.clone()                                                            # :77
    loss_out = torch.empty(B, device=x.device, dtype=torch.float32)
    pred_out = torch.empty(B, device=x.device, dtype=torch.uint8)
    grad = _apgd._model_fwd_bwd(model, start, y_hard, y_soft, None, loss_out, pred_out, True, 0, None, sign_ok=True)   # :88-93
    out = torch.empty_like(x)
    _lib.check(lib.apgd_fgsm_step_f32(x.data_ptr(), start.data_ptr(), grad.data_ptr(), _lib.dtype_code(grad.dtype), out.data_ptr(),
                                      x.numel(), float(alpha * eps), eps, project, stream), "apgd_fgsm_step_f32")       # :95-98
    return out
