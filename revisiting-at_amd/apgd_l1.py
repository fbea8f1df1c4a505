"""The L1 branch of ``apgd_train`` (``/root/reference/autopgd_train_clean.py:160-167, 239-250, 351-362``, ``L1_projection`` ``:24-91``)
on the device.  SURVEY.md marks this branch "next" (it is not on the benchmark's AT path; the reference's own code only takes
NCHW-contiguous inputs, ``:240``), so it is built for parity first: the per-sample order statistics, sorts and prefix sums of the
sparse step and of the projection run in the ROCm libraries behind ``torch.sort`` / ``cumsum`` on device tensors, forward / loss /
prediction / best-point tracking are the HIP kernels of the Linf path (``apgd_loss_pred``, ``apgd_state_update``,
``apgd_track_rows``), and nothing synchronises with the host (the reference's ``nonzero()`` selections are masks here).
Agreement with the oracle: fp32 summation-order noise, like the L2 branch."""
from __future__ import annotations

import math

import torch

from . import _lib


def l1_projection(x: torch.Tensor, y: torch.Tensor, eps: float) -> torch.Tensor:
    """``:24-91`` for ``[B, n]`` fp32 device tensors: delta with ``||y + delta||_1 <= eps`` and ``0 <= x + y + delta <= 1``."""
    B, n = x.shape
    sigma = torch.sign(y)
    u = torch.minimum((1.0 - x) - y, x + y).clamp_(max=0.0)
    l = -y.abs()
    neg = torch.cat((u, l), 1).neg_()
    bs, indbs = torch.sort(neg, dim=1, stable=True)
    del neg
    bs2 = torch.cat((bs[:, 1:], bs.new_zeros(B, 1)), 1)
    size1 = (indbs < n).to(torch.float32).mul_(2.0).sub_(1.0).cumsum_(1)
    del indbs
    s1 = -u.sum(1)
    c = eps - y.abs().sum(1)
    need = (s1 + c) < 0                                               # samples outside the ball (c5, :52)
    s = (bs2 - bs).mul_(size1).cumsum_(1).add_(s1[:, None])
    del bs
    lb = torch.zeros(B, device=x.device, dtype=torch.float32)
    ub = torch.full((B,), float(2 * n - 1), device=x.device, dtype=torch.float32)
    for _ in range(int(math.ceil(math.log2(2 * n)))):                 # :66-81, for every sample (masked below)
        mid = torch.floor((lb + ub) / 2.0)
        c8 = (s.gather(1, mid.long()[:, None]).squeeze(1) + c) < 0
        lb = torch.where(c8, mid, lb)
        ub = torch.where(c8, ub, mid)
    lb2 = lb.long()[:, None]
    nxt = (lb2 + 1).clamp_(max=2 * n - 1)
    alpha = (-s.gather(1, lb2).squeeze(1) - c) / size1.gather(1, nxt).squeeze(1) + bs2.gather(1, lb2).squeeze(1)
    d = torch.where(need[:, None], -torch.minimum(torch.maximum(-u, alpha[:, None]), -l), u)
    return sigma * d


def l1_step(x, x_adv, grad, step_size, topk, eps):
    """``:239-250`` on ``[B, n]`` rows: sparse signed step, then the projection."""
    B, n = x.shape
    ag = grad.abs()
    srt = torch.sort(ag, dim=1)[0]
    pos = ((1.0 - topk) * float(n)).clamp_(0, n - 1).long()
    thr = srt.gather(1, pos[:, None])
    del srt
    sg = torch.sign(grad * (ag >= thr).to(grad.dtype))
    cnt = sg.abs().sum(1, keepdim=True) + 1e-10
    x1 = x_adv + (step_size[:, None] * sg) / cnt
    delta_u = x1 - x
    return (x + delta_u) + l1_projection(x, delta_u, eps)


def apgd_l1(model, x, y, eps, n_iter, soft, is_train, verbose, fwd_bwd, ApgdWorkspace, stream_ptr):
    """The L1 attack loop; returns the reference's tuple.  ``fwd_bwd`` / ``ApgdWorkspace`` / ``stream_ptr`` come from ``apgd.py``
    (one forward / loss / backward of the model; per-call buffers; the current HIP stream)."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise _lib.ApgdHipError("apgd_train needs a device (MI355X) tensor; there is no CPU fallback")
    if x.dtype != torch.float32:
        raise _lib.ApgdHipError(f"attack state is fp32 (got {x.dtype})")
    lib = _lib.load()
    n_iter = int(n_iter)
    # the L1 branch works on NCHW-contiguous iterates (the reference's own :240 reshapes them); a channels_last input gets its
    # results back in its own memory format, as apgd_train promises
    cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    x = x.detach().contiguous()
    B = x.shape[0]
    E = x[0].numel() if B > 0 else 0
    stream = stream_ptr()
    if soft:
        y_soft, y_hard = y.detach().to(torch.float32).contiguous(), None
    else:
        y_hard, y_soft = y.detach().to(torch.int64).contiguous(), None
    ws = ApgdWorkspace(x, n_iter, n_rot=1)
    cur = ws.rot[0]
    _lib.check(lib.apgd_init_f32(x.data_ptr(), cur.data_ptr(), ws.x_best.data_ptr(), ws.x_best_adv.data_ptr(), x.numel(), stream),
               "apgd_init_f32")                                        # :135, 141-143
    k = max(int(.04 * n_iter), 1)                                      # :161
    topk = torch.full((B,), .05 if is_train else .2, device=x.device, dtype=torch.float32)   # :162-163
    sp_old = torch.full((B,), float(E), device=x.device, dtype=torch.float32)                # :164
    adasp_redstep, adasp_minstep, alpha = 1.5, 10., 1.                 # :165-167
    step_size = torch.full((B,), alpha * eps, device=x.device, dtype=torch.float32)          # :169-170
    counter3 = 0
    grad = fwd_bwd(model, cur, y_hard, y_soft, ws, ws.loss_best, ws.acc, True, 0, None, False)   # :174-200 (fp32 gradient: its VALUES rank)
    grad = grad.contiguous() if not grad.is_contiguous() else grad
    grad_best = grad.clone()
    xf = x.view(B, -1)
    for i in range(n_iter):                                            # :209
        new = l1_step(xf, cur.view(B, -1), grad.reshape(B, -1), step_size, topk, float(eps)).view_as(x)
        cur = new.contiguous()
        last = i == n_iter - 1
        g_new = fwd_bwd(model, cur, y_hard, y_soft, ws, ws.loss, ws.pred, not last, 0, None, False)   # :266-287
        if g_new is not None:
            grad = g_new.contiguous() if not g_new.is_contiguous() else g_new
        # acc, loss_steps, best-loss bookkeeping and the row moves of :296-324 (no step-size check for L1: do_check = 0)
        _lib.check(lib.apgd_state_update(ws.loss.data_ptr(), ws.pred.data_ptr(), ws.acc.data_ptr(), ws.loss_best.data_ptr(),
                                         ws.loss_best_last.data_ptr(), ws.reduced_last.data_ptr(), step_size.data_ptr(),
                                         ws.loss_steps.data_ptr(), ws.flags.data_ptr(), B, max(n_iter, 1), i, 0, 1, 0.75, stream),
                   "apgd_state_update")
        _lib.check(lib.apgd_track_rows(ws.flags.data_ptr(), cur.data_ptr(), grad.data_ptr(), ws.x_best.data_ptr(), grad_best.data_ptr(),
                                       ws.x_best_adv.data_ptr(), grad.element_size(), B, E, int(last), stream),
                   "apgd_track_rows")                                  # (final: the dead grad_best copy of the last iteration is skipped)
        counter3 += 1
        if counter3 == k:                                              # :351-362: adapt the sparsity
            sp_curr = (ws.x_best != x).view(B, -1).sum(1).to(torch.float32)
            red = (sp_curr / sp_old) < .95
            topk = sp_curr / float(E) / 1.5
            step_size = torch.where(red, torch.full_like(step_size, alpha * eps), step_size / adasp_redstep)
            step_size.clamp_(alpha * eps / adasp_minstep, alpha * eps)
            sp_old = sp_curr
            shp = (B,) + (1,) * (x.dim() - 1)
            cur = torch.where(red.view(shp), ws.x_best, cur)
            grad = torch.where(red.view(shp), grad_best, grad)
            counter3 = 0
        if verbose:
            print('iteration: {} - best loss: {:.6f} - robust accuracy: {:.2%} - step size: {:.5f} - topk: {:.2f}'.format(
                i, ws.loss_best.sum().item(), ws.acc.float().mean().item(), step_size.mean().item(), (topk.mean() * E).item()))
    xb, xba = ws.x_best, ws.x_best_adv
    if cl:
        xb, xba = xb.contiguous(memory_format=torch.channels_last), xba.contiguous(memory_format=torch.channels_last)
    return xb, ws.acc.view(torch.bool), ws.loss_best, xba
