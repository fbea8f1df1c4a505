"""Model-side operators of the hot path.

Each operator has ONE product implementation: hand-written gfx950 kernels
(``csrc/model_kernels.hip`` via ``include/convnext_hip.h``) for the bandwidth-bound pieces the
ROCm libraries handle badly (depthwise 7x7 in NHWC, LayerNorm(+GELU) over channels-last rows),
PyTorch-ROCm library calls (hipBLASLt / MIOpen) for plain GEMMs and dense convolutions.  Device
tensors always take the HIP path and raise if the library is missing; there is no CPU path for
the fused operators.  ``APGD_OPS=eager`` swaps the fused operators for their library-call
composition, for A/B timing only (bench.py records the mode).

Activation dtype: under ``torch.autocast`` the fused operators emit the autocast dtype (bf16),
with fp32 accumulation and fp32 statistics inside the kernels; outside autocast they emit fp32.
The residual stream stays fp32 exactly as in the reference's autocast run (``gamma`` is fp32).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _lib

MODE = os.environ.get("APGD_OPS", "hip")

# Set (process-wide, read at backward time from the autograd engine's thread) while the attack asks
# for the input gradient only (autopgd_train_clean.py:185, 283: autograd.grad(loss, [x_adv])).
# torch's own ops get this from the engine's per-call output mask; Python Functions only see the
# static ctx.needs_input_grad, so the fused operators consult this flag to skip parameter gradients.
_INPUT_GRAD_ONLY = False


class input_grad_only:
    """Context manager used by apgd_train around its ``torch.autograd.grad`` call."""

    def __enter__(self):
        global _INPUT_GRAD_ONLY
        self._prev = _INPUT_GRAD_ONLY
        _INPUT_GRAD_ONLY = True
        return self

    def __exit__(self, *exc):
        global _INPUT_GRAD_ONLY
        _INPUT_GRAD_ONLY = self._prev
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _act_dtype(x):
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda')
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def _code(t):
    return _lib.dtype_code(t.dtype)


def _rows(x_nchw):
    """[N,C,H,W] (any memory format) -> contiguous [N,H,W,C] view/copy."""
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _f32(p):
    p = p.detach()
    return p if p.dtype == torch.float32 else p.float()


# ------------------------------------------------------------------------------ LayerNorm (+GELU) over rows
class _LayerNormRows(torch.autograd.Function):
    """y = [GELU](LN(x)) over the last dim of a contiguous [..., C] tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, gelu, out_dtype):
        lib = _lib.load()
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        w, b = _f32(weight), _f32(bias)
        _lib.check(lib.cnx_layernorm_fwd(x.data_ptr(), _code(x), w.data_ptr(), b.data_ptr(), eps, y.data_ptr(),
                                         _code(y), mean.data_ptr(), rstd.data_ptr(), M, C, int(gelu), _stream()),
                   "cnx_layernorm_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.gelu, ctx.C, ctx.M = gelu, C, M
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, b, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C, M = ctx.C, ctx.M
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
        want_p = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY
        dw = db = ws = None
        if want_p:
            dw = torch.empty(C, device=x.device, dtype=torch.float32)
            db = torch.empty(C, device=x.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_layernorm_bwd(dy.data_ptr(), _code(dy), x.data_ptr(), _code(x), w.data_ptr(), b.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _code(dx), _lib.ptr(dw),
                                         _lib.ptr(db), _lib.ptr(ws), M, C, int(ctx.gelu), _stream()),
                   "cnx_layernorm_bwd")
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None, None


def _ln_rows(x_rows, weight, bias, eps, gelu):
    if not x_rows.is_cuda:
        raise _lib.ApgdHipError("fused LayerNorm needs a device tensor (no CPU path in the product)")
    if x_rows.dtype not in (torch.float32, torch.bfloat16):
        x_rows = x_rows.float()
    return _LayerNormRows.apply(x_rows, weight, bias, float(eps), bool(gelu), _act_dtype(x_rows))


def layer_norm_cf(x, weight, bias, eps):
    """LayerNorm over dim 1 of ``[N,C,H,W]`` (``utils_architecture.py:76-81``; timm LayerNorm2d)."""
    if MODE == "eager":
        return F.layer_norm(x.permute(0, 2, 3, 1), weight.shape, weight, bias, eps).permute(0, 3, 1, 2)
    return _ln_rows(_rows(x), weight, bias, eps, False).permute(0, 3, 1, 2)


def layer_norm_cf_gelu(x, weight, bias, eps):
    """``GELU(LN_cf(x))`` — the ConvStem pair (``utils_architecture.py:128-129`` etc.), one kernel."""
    if MODE == "eager":
        return F.gelu(F.layer_norm(x.permute(0, 2, 3, 1), weight.shape, weight, bias, eps).permute(0, 3, 1, 2))
    return _ln_rows(_rows(x), weight, bias, eps, True).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ depthwise 7x7 + LayerNorm
class _DwConvLN(torch.autograd.Function):
    """[N,H,W,C] rows in -> LN(dwconv7x7(x)) rows out (``models/convnext.py:39-41``)."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_w, ln_b, eps, out_dtype):
        lib = _lib.load()
        N, H, W, C = x.shape
        w49c = _f32(dw_w).reshape(C, 49).t().contiguous()            # [49][C] tap-major
        dwb = _f32(dw_b) if dw_b is not None else None
        lw, lb = _f32(ln_w), _f32(ln_b)
        dwo = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        _lib.check(lib.cnx_dwconv7x7_nhwc(x.data_ptr(), _code(x), w49c.data_ptr(), _lib.ptr(dwb), None, dwo.data_ptr(),
                                          _code(dwo), N, H, W, C, 0, _stream()), "cnx_dwconv7x7_nhwc")
        M = N * H * W
        y = torch.empty_like(dwo)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_layernorm_fwd(dwo.data_ptr(), _code(dwo), lw.data_ptr(), lb.data_ptr(), eps, y.data_ptr(),
                                         _code(y), mean.data_ptr(), rstd.data_ptr(), M, C, 0, _stream()),
                   "cnx_layernorm_fwd")
        ctx.save_for_backward(x, w49c, dwo, mean, rstd, lw)
        ctx.has_bias = dw_b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w49c, dwo, mean, rstd, lw = ctx.saved_tensors
        N, H, W, C = x.shape
        M = N * H * W
        dy = dy.contiguous()
        need_ln_p = (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) and not _INPUT_GRAD_ONLY
        need_dw_p = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY
        d_dwo = torch.empty_like(dwo)
        dlw = dlb = ws = None
        if need_ln_p:
            dlw = torch.empty(C, device=x.device, dtype=torch.float32)
            dlb = torch.empty(C, device=x.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_layernorm_bwd(dy.data_ptr(), _code(dy), dwo.data_ptr(), _code(dwo), lw.data_ptr(), None,
                                         mean.data_ptr(), rstd.data_ptr(), d_dwo.data_ptr(), _code(d_dwo),
                                         _lib.ptr(dlw), _lib.ptr(dlb), _lib.ptr(ws), M, C, 0, _stream()),
                   "cnx_layernorm_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.cnx_dwconv7x7_nhwc(d_dwo.data_ptr(), _code(d_dwo), w49c.data_ptr(), None, None, dx.data_ptr(),
                                              _code(dx), N, H, W, C, 1, _stream()), "cnx_dwconv7x7_nhwc(flip)")
        dww = dwb = None
        if need_dw_p:
            g49 = torch.empty(49, C, device=x.device, dtype=torch.float32)
            dwb = torch.empty(C, device=x.device, dtype=torch.float32)
            ws2 = torch.empty(lib.cnx_dwconv7x7_wgrad_ws_floats(C), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_dwconv7x7_wgrad_nhwc(x.data_ptr(), _code(x), d_dwo.data_ptr(), _code(d_dwo),
                                                    g49.data_ptr(), dwb.data_ptr(), ws2.data_ptr(), N, H, W, C,
                                                    _stream()), "cnx_dwconv7x7_wgrad_nhwc")
            dww = g49.t().reshape(C, 1, 7, 7)
            if not ctx.has_bias:
                dwb = None
        return dx, dww, dwb, dlw, dlb, None, None


def dwconv_ln(x_rows, dw_w, dw_b, ln_w, ln_b, eps):
    """``LN(dw7x7(x))`` on contiguous ``[N,H,W,C]`` rows (fp32 or bf16) -> rows in the activation dtype."""
    if not x_rows.is_cuda:
        raise _lib.ApgdHipError("fused depthwise-7x7+LayerNorm needs a device tensor (no CPU path in the product)")
    if x_rows.dtype not in (torch.float32, torch.bfloat16):
        x_rows = x_rows.float()
    return _DwConvLN.apply(x_rows, dw_w, dw_b, ln_w, ln_b, float(eps), _act_dtype(x_rows))


# ------------------------------------------------------------------------------ fused MLP (MFMA)
def _w2_perm(n_hidden, device):
    """Hidden-index order the MFMA accumulators enumerate (see csrc/mlp_kernels.hip / convnext_hip.h)."""
    p = torch.arange(32)
    t, half, e = p // 16, (p // 8) % 2, p % 8
    src = (e & 3) + 8 * (2 * t + (e >> 2)) + 4 * half
    return (torch.arange(0, n_hidden, 32).view(-1, 1) + src.view(1, -1)).reshape(-1).to(device)


_wcache = {}


def _cached(param, tag, fn):
    """bf16 / permuted copies of a parameter, rebuilt only when the parameter changes
    (optimizer steps bump ``_version``); the attack's 3 forwards and the train forward share them."""
    key = (id(param), tag)
    ver = (param._version, param.data_ptr())
    hit = _wcache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    with torch.no_grad():
        val = fn(param.detach())
    _wcache[key] = (ver, val)
    return val


class _MlpFused(torch.autograd.Function):
    """rows a [.., C] (bf16) , residual x [.., C] -> x + gamma * fc2(GELU(fc1(a)))  (``models/convnext.py:42-49``).

    Forward: one MFMA kernel (``cnx_mlp_fwd``), hidden activation stays on-chip.  Backward recomputes the
    hidden pre-activation instead of storing it (M x 4C bf16 per block is the largest tensor of the model)."""

    @staticmethod
    def forward(ctx, a, x, w1, b1, w2, b2, gamma):
        lib = _lib.load()
        C = a.shape[-1]
        M = a.numel() // C
        w1b = _cached(w1, "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        w2b = _cached(w2, "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        w2p = _cached(w2, "perm", lambda w: w.to(torch.bfloat16)[:, _w2_perm(w.shape[1], w.device)].contiguous())
        b1f, b2f = _f32(b1), _f32(b2)
        gf = _f32(gamma) if gamma is not None else None
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32 if (gamma is not None or x.dtype == torch.float32)
                          else x.dtype)
        need_p = any(ctx.needs_input_grad[2:])
        y2 = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16) if (need_p and gamma is not None
                                                                              and torch.is_grad_enabled()) else None
        _lib.check(lib.cnx_mlp_fwd(a.data_ptr(), w1b.data_ptr(), b1f.data_ptr(), w2p.data_ptr(), b2f.data_ptr(),
                                   _lib.ptr(gf), x.data_ptr(), _code(x), out.data_ptr(), _code(out), _lib.ptr(y2), M, C,
                                   _stream()), "cnx_mlp_fwd")
        ctx.save_for_backward(a, w1b, w2b, b1f, b2f, gf, y2)
        ctx.x_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        a, w1b, w2b, b1f, b2f, gf, y2 = ctx.saved_tensors
        C = a.shape[-1]
        g2 = g.reshape(-1, C)
        a2 = a.reshape(-1, C)
        dos = (g2 * gf if gf is not None else g2).to(torch.bfloat16)            # d(fc2 out)
        hpre = torch.addmm(b1f.to(torch.bfloat16), a2, w1b.t())                  # recomputed, [M, 4C]
        dh = dos @ w2b                                                           # [M, 4C]
        dhpre = torch.ops.aten.gelu_backward(dh, hpre)
        da = (dhpre @ w1b).view_as(a) if ctx.needs_input_grad[0] else None
        dx = g.to(ctx.x_dtype) if ctx.needs_input_grad[1] else None
        dw1 = db1 = dw2 = db2 = dgamma = None
        if any(ctx.needs_input_grad[2:]) and not _INPUT_GRAD_ONLY:
            h = F.gelu(hpre)
            dw2 = (dos.t() @ h).float()
            db2 = dos.float().sum(0)
            dw1 = (dhpre.t() @ a2).float()
            db1 = dhpre.float().sum(0)
            if gf is not None:
                y2v = y2.reshape(-1, C) if y2 is not None else torch.addmm(b2f.to(torch.bfloat16), h, w2b.t())
                dgamma = (g2.float() * y2v.float()).sum(0)
        return da, dx, dw1, db1, dw2, db2, dgamma


def mlp_fused_supported(C):
    return bool(_lib.load().cnx_mlp_fwd_supported(C))


# Widths for which the fused MFMA MLP forward is switched on.  Measured on MI355X (tools/mlp_bench.py,
# B=256): it beats the hipBLASLt + aten composition only where the block is HBM-bound (C=96); at
# C>=192 the in-register GELU makes it VALU-bound and the library GEMMs win, so those stay on
# hipBLASLt until the kernel is tuned.  APGD_MLP_FUSED="96,192,384" / "" overrides.
_FUSED_MLP_WIDTHS = {int(v) for v in os.environ.get("APGD_MLP_FUSED", "").split(",") if v.strip()}


def _use_fused_mlp(C):
    return MODE != "eager" and C in _FUSED_MLP_WIDTHS and mlp_fused_supported(C)


def convnext_block(x, dw_w, dw_b, ln_w, ln_b, eps, w1, b1, w2, b2, gamma):
    """``x + gamma * fc2(GELU(fc1(LN(dw7x7(x)))))`` on ``[N,C,H,W]`` (``models/convnext.py:37-50``)."""
    if MODE == "eager":
        y = F.conv2d(x, dw_w, dw_b, padding=3, groups=x.shape[1]).permute(0, 2, 3, 1)
        y = F.layer_norm(y, ln_w.shape, ln_w, ln_b, eps)
    else:
        xr = _rows(x)
        y = dwconv_ln(xr, dw_w, dw_b, ln_w, ln_b, eps)
        if _use_fused_mlp(x.shape[1]) and y.dtype == torch.bfloat16:
            return _MlpFused.apply(y, xr, w1, b1, w2, b2, gamma).permute(0, 3, 1, 2)
    y = F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    if gamma is not None:
        y = y * gamma
    return x + y.permute(0, 3, 1, 2)


def attention(qkv, num_heads, scale):
    """Multi-head softmax attention from a packed ``[B,N,3C]`` projection -> ``[B,N,C]``
    (timm 0.8 ``Attention.forward``)."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).unbind(0)
    a = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, C)
