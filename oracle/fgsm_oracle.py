"""TEST INFRASTRUCTURE - numpy-fp32 restatement of the reference's ``fgsm_train`` (``/root/reference/fgsm_train.py:72-100``), the
other value of the attack selector that sits behind ``WrappedModel`` (``main.py:836-842``: ``partial(fgsm_train, eps, use_rs=True,
alpha, noise_level, skip_projection)``).  Pinned by ``tests/golden/fgsm_*.npz`` (recorded from the reference, bit for bit).

Every operation is rounded to fp32 on its own, Python scalars enter as fp32 (what torch does with a float tensor and a Python
float); ``alpha * eps`` is formed in double first because the reference writes ``alpha * eps * grad.sign()`` (``:95``).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def sign_f32(g: np.ndarray) -> np.ndarray:
    """torch.sign: -1 / 0 / +1, sign(NaN) = 0 on the comparison form the kernels use (never met: fixtures are finite)."""
    return (g > 0).astype(F32) - (g < 0).astype(F32)


def fgsm_start(x: np.ndarray, t, eps: float, use_rs: bool, noise_level: float, skip_projection: bool) -> np.ndarray:
    """``:76-84``: the clean point, or ``x + (2 t - 1) * eps * noise_level`` (clamped to [0, 1] unless skip_projection)."""
    x = x.astype(F32)
    if not use_rs:
        return x.copy()
    v = (F32(2.0) * t.astype(F32) - F32(1.0)) * F32(eps)
    v = v * F32(noise_level)
    x_adv = x + v
    if not skip_projection:
        x_adv = np.minimum(np.maximum(x_adv, F32(0.0)), F32(1.0))
    return x_adv.astype(F32)


def fgsm_step(x: np.ndarray, x_adv: np.ndarray, grad: np.ndarray, eps: float, alpha: float, skip_projection: bool) -> np.ndarray:
    """``:95-98``: one signed step of size alpha * eps, projection onto the eps-ball around x and onto [0, 1]."""
    out = x_adv.astype(F32) + F32(alpha * eps) * sign_f32(np.asarray(grad, dtype=F32))
    if not skip_projection:
        d = np.minimum(np.maximum(out - x.astype(F32), F32(-eps)), F32(eps))
        out = x.astype(F32) + d
        out = np.minimum(np.maximum(out, F32(0.0)), F32(1.0))
    return out.astype(F32)


def fgsm_train_oracle(fwd_bwd, x: np.ndarray, y, eps: float, t=None, alpha: float = 1.25, use_rs: bool = False,
                      noise_level: float = 1.0, skip_projection: bool = False):
    """``fwd_bwd(x_adv, need_grad=True) -> (logits, grad, _)`` as in ``apgd_oracle`` (a ``ReplayModel`` or a ``TorchModelAdapter``).
    Returns (x_adv_out, x_fed): the attack's result and the iterate the model was evaluated at."""
    if use_rs and t is None:
        raise ValueError("use_rs needs the uniform draw t (torch.rand_like(x) in the reference)")
    x_fed = fgsm_start(x, t, eps, use_rs, noise_level, skip_projection)
    _, grad, _ = fwd_bwd(x_fed, True)
    return fgsm_step(x, x_fed, grad, eps, alpha, skip_projection), x_fed
