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

import contextlib
import math
import os
import weakref

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


# Set by apgd_train around a model FORWARD whose backward will ask for the input gradient only (the attack's own calls):
# lets an operator pick a forward variant that saves what only an input-gradient backward can use (the C = 384 block's Hpre
# workspace) - at forward time a Python autograd.Function cannot know which gradients the engine will request.
_ATTACK_FWD = False


class attack_forward:
    def __enter__(self):
        global _ATTACK_FWD
        self._prev = _ATTACK_FWD
        _ATTACK_FWD = True
        return self

    def __exit__(self, *exc):
        global _ATTACK_FWD
        _ATTACK_FWD = self._prev
        return False


# Set by apgd._model_fwd_bwd around every model call of the attack on a model whose captured attack runs as batch chunks on
# SEVERAL streams (apgd.two_stream_model: the narrow ConvNeXt pyramids).  Under APGD_GEMM=auto those calls' pointwise convolutions /
# linears all run on cnx_gemm_nt: two of the library's stream-K GEMMs in flight at once deadlocked the GPU (ViT-B shapes,
# hipBLASLt ``..._SK3_...`` solutions spinning on a shared synchroniser), so overlapping chunks keep the library's GEMMs out of
# their streams - and the eager attack of the same model takes the same kernels, so that replay and eager loop stay bit-identical.
_ATTACK_PASS = False


class attack_pass:
    def __enter__(self):
        global _ATTACK_PASS
        self._prev = _ATTACK_PASS
        _ATTACK_PASS = True
        return self

    def __exit__(self, *exc):
        global _ATTACK_PASS
        _ATTACK_PASS = self._prev
        return False


# Gradient-sign sink of the Linf attack.  ``step_size * sign(grad)`` (autopgd_train_clean.py:221) is the only use the Linf
# update makes of the input gradient, so when the layer that produces it is our own first stem convolution AND its input
# is the attack iterate itself, the attack hands it an int8 buffer: the kernel stores sign(dx) (a quarter of the bytes)
# taken from the same fp32 accumulator it would have stored, and autograd gets a stride-0 zero in place of dx.
_SIGN_SINK = None


class grad_sign_sink:
    """``with grad_sign_sink(x_in) as sink: autograd.grad(...)`` - afterwards ``sink.signs`` is the int8 sign tensor
    (same shape / layout as ``x_in``) if the stem kernel produced it, else None (the caller uses autograd's gradient)."""

    def __init__(self, x_in, blocked=False):
        self.x_in, self.signs, self._buf = x_in, None, None
        # blocked=True (the Linf APGD loop asks for it): the sign order of include/apgd_hip.h, APGD_I8_BLK - the update kernel then
        # reads 16 signs per lane in one load.  Whole 1024-element groups per sample only; ``signs.apgd_blocked`` tells the
        # attack which dtype code to pass.  Other users of the sink (fgsm) get element order.
        self.blocked = bool(blocked) and SIGN_BLOCKED and x_in.dim() >= 2 and x_in.shape[0] > 0 and x_in[0].numel() % 1024 == 0

    def buffer(self):
        if self._buf is None:
            self._buf = torch.empty(self.x_in.shape, device=self.x_in.device, dtype=torch.int8)
            self._buf.apgd_blocked = self.blocked
        return self._buf

    def __enter__(self):
        global _SIGN_SINK
        self._prev = _SIGN_SINK
        _SIGN_SINK = self
        return self

    def __exit__(self, *exc):
        global _SIGN_SINK
        _SIGN_SINK = self._prev
        return False


# Measured (tools/k1_ab.py, B = 256 x 3 x 224 x 224, alternating launches): the blocked order is 2.5 % SLOWER than element order (119.6 vs
# 116.6 us warm, 129.6 vs 127.4 us behind a 512 MB copy) - the int8 stream's request count was not what holds the update kernel at
# 0.64 - 0.70 of 8 TB/s of moved bytes - so it is off (a module constant since round 4; the kernels stay, the tests switch it on by attribute).
SIGN_BLOCKED = False


def signs_to_linear(t):
    """Blocked int8 signs (``cnx_stem_conv_dgrad_sign_blk``) -> element order; a tensor without the ``apgd_blocked`` mark is
    returned as it is.  For tests / replay: element g*1024 + (u*64 + l)*4 + j is stored at byte g*1024 + l*16 + u*4 + j."""
    if not getattr(t, "apgd_blocked", False):
        return t
    B = t.shape[0]
    return t.reshape(B, -1, 64, 4, 4).permute(0, 1, 3, 2, 4).reshape(t.shape)


def signs_to_blocked(t):
    """Inverse of ``signs_to_linear`` for a linear int8 tensor [B, ...] with numel per sample % 1024 == 0 (tests)."""
    B = t.shape[0]
    out = t.reshape(B, -1, 4, 64, 4).permute(0, 1, 3, 2, 4).reshape(t.shape).contiguous()
    out.apgd_blocked = True
    return out


def _sink_for(x):
    """The active sink if ``x`` (the stem convolution's saved input) is the attack iterate it was opened for."""
    sk = _SIGN_SINK
    if (sk is not None and _INPUT_GRAD_ONLY and x.data_ptr() == sk.x_in.data_ptr() and x.shape == sk.x_in.shape
            and x.is_contiguous() and sk.x_in.is_contiguous()):
        return sk
    return None


def _stem_dgrad(lib, x, gr, wq, N, H, W, P):
    """Input gradient of the first stem convolution: int8 signs into the attack's sink, or fp32 dx."""
    sk = _sink_for(x)
    if sk is not None:
        buf = sk.buffer()
        fn = lib.cnx_stem_conv_dgrad_sign_blk if sk.blocked else lib.cnx_stem_conv_dgrad_sign
        _lib.check(fn(gr.data_ptr(), wq.data_ptr(), buf.data_ptr(), N, H, W, P, _stream()), "cnx_stem_conv_dgrad_sign")
        sk.signs = buf
        return torch.zeros((), device=x.device, dtype=x.dtype).expand(x.shape)
    dx = torch.empty_like(x)
    _lib.check(lib.cnx_stem_conv_dgrad(gr.data_ptr(), wq.data_ptr(), dx.data_ptr(), N, H, W, P, _stream()),
               "cnx_stem_conv_dgrad")
    return dx


def _stream():
    return _SIDE.handle if _SIDE is not None else torch.cuda.current_stream().cuda_stream


# The weight-gradient work of a block's backward - the paired contraction, its fixed-order sum, the per-channel identities: matrix-pipe /
# ingest bound, nothing downstream waits for it - runs on a SIDE stream under the depthwise stencil's input- and filter-gradient kernels
# of the same backward call (HBM / VALU bound), joined before the call returns (round 6).  Every tensor stays a tensor of the CURRENT
# stream for the allocator: only the launches of our own entry points move (``_stream()``), no torch kernel may run inside the context,
# and temporaries made inside are kept alive until the join (``_keep``) - freed earlier, the current stream could be handed their memory
# while the side stream still uses it.
_SIDE = None
_WGRAD_SIDE = os.environ.get("APGD_WGRAD_SIDE", "1") != "0"
_side_streams = {}


class _SideLaunch:
    def __init__(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = _side_streams.get(idx)
        if st is None:
            st = _side_streams[idx] = torch.cuda.Stream(device=idx)
        self.stream, self.handle, self.kept = st, st.cuda_stream, []

    def __enter__(self):
        global _SIDE
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        _SIDE = self
        return self

    def __exit__(self, *exc):
        global _SIDE
        _SIDE = None
        return False

    def join(self):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        torch.cuda.current_stream().wait_event(ev)
        self.kept.clear()


def _keep(*tensors):
    if _SIDE is not None:
        _SIDE.kept.extend(t for t in tensors if t is not None)


def _act_dtype(x):
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda')
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def _hip_dtype_ok(x):
    """The hand-written operators compute in fp32 / bf16.  Under any other activation dtype - e.g. the reference's own
    ``autocast(enabled=True)``, which is fp16 (``main.py:985``) - the operator dispatchers below run the plain library
    composition instead of failing inside a kernel with APGD_ERR_DTYPE."""
    return _act_dtype(x) in (torch.float32, torch.bfloat16)


def _code(t):
    return _lib.dtype_code(t.dtype)


def _rows(x_nchw):
    """[N,C,H,W] (any memory format) -> contiguous [N,H,W,C] view/copy."""
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _f32(p):
    p = p.detach()
    return p if p.dtype == torch.float32 else p.float()


# ------------------------------------------------------------------------------ global average pool over channels-last rows
class _GlobalPoolRows(torch.autograd.Function):
    """``x.mean((-2, -1), keepdim=True)`` (the head's global pool, timm ``SelectAdaptivePool2d('avg')``) of an NCHW-shaped view of
    channels-last rows.  autograd's own backward materialises the gradient NCHW-contiguous ([N, C, H, W] / HW) and the last block's
    backward then copies it into rows: two strided element-wise kernels (55 us each at 7 x 7 x 768, batch 256).  Here the gradient is
    written once, in rows."""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        ctx.shape = (N, C, H, W)
        rows = x.permute(0, 2, 3, 1)                                         # [N, H, W, C], contiguous by the caller's check
        return rows.reshape(N, H * W, C).mean(1).view(N, C, 1, 1)

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.shape
        gr = (g.reshape(N, 1, C) * (1.0 / (H * W))).expand(N, H * W, C).contiguous()
        return gr.view(N, H, W, C).permute(0, 3, 1, 2)


_POOL_ROWS = os.environ.get("APGD_POOL_ROWS", "1") != "0"


def global_pool(x):
    """[N, C, H, W] -> [N, C, 1, 1] mean over the map."""
    if _POOL_ROWS and MODE != "eager" and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.permute(0, 2, 3, 1).is_contiguous():
        return _GlobalPoolRows.apply(x)
    return x.mean((-2, -1), keepdim=True)


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


class _LayerNormRowsSkip(torch.autograd.Function):
    """``(LN(x), x)`` for an fp32 residual stream ``x`` that feeds both a LayerNorm branch and the skip connection around it
    (timm ``Block.forward``: ``x + ls(attn(norm1(x)))``).  Autograd hands BOTH output gradients to one backward call, so the
    sum ``d(skip) + LN'(d(branch))`` is formed inside the LayerNorm backward kernel (``cnx_layernorm_bwd_add``) instead of by a
    separate pass over the residual stream (two 155 MB fp32 reads and one write per block and backward pass at ViT-B / 224)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        lib = _lib.load()
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        w, b = _f32(weight), _f32(bias)
        _lib.check(lib.cnx_layernorm_fwd(x.data_ptr(), _code(x), w.data_ptr(), b.data_ptr(), eps, y.data_ptr(), _code(y),
                                         mean.data_ptr(), rstd.data_ptr(), M, C, 0, _stream()), "cnx_layernorm_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.C, ctx.M = C, M
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        lib = _lib.load()
        x, w, b, mean, rstd = ctx.saved_tensors
        C, M = ctx.C, ctx.M
        want_p = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY
        if dy is None:                                   # the LayerNorm output was not used
            return dskip, None, None, None, None
        dy = dy.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous()
            if dskip.dtype != torch.float32:
                dskip = dskip.float()
        dx = torch.empty_like(x)
        dw = db = ws = None
        if want_p:
            dw = torch.empty(C, device=x.device, dtype=torch.float32)
            db = torch.empty(C, device=x.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_layernorm_bwd_add(dy.data_ptr(), _code(dy), x.data_ptr(), _code(x), w.data_ptr(), b.data_ptr(),
                                             mean.data_ptr(), rstd.data_ptr(), _lib.ptr(dskip), dx.data_ptr(), _code(dx),
                                             _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), M, C, 0, _stream()),
                   "cnx_layernorm_bwd_add")
        return dx, dw, db, None, None


_LN_SKIP = True


def layer_norm_skip(x, weight, bias, eps):
    """``(LN(x), x')`` where ``x'`` is ``x`` for the skip connection around the LayerNorm branch: use ``x'`` in the residual sum
    and the backward adds the two gradients of ``x`` inside the LayerNorm backward kernel.  Falls back to the plain pair."""
    if (MODE == "eager" or not _LN_SKIP or not x.is_cuda or x.dtype != torch.float32 or x.shape[-1] % 4 != 0 or not _hip_dtype_ok(x)
            or not x.is_contiguous() or not (torch.is_grad_enabled() and x.requires_grad)):
        return layer_norm_last(x, weight, bias, eps), x
    return _LayerNormRowsSkip.apply(x, weight, bias, float(eps), _act_dtype(x))


class _ScaleResidual(torch.autograd.Function):
    """``x + gamma * y`` over rows (``x`` fp32 or bf16 residual stream, ``y`` bf16 branch output, ``gamma`` [C] or None) ->
    fp32: the residual connections of the ViT blocks (timm ``Block.forward``: ``x + ls(attn(norm(x)))``).  One kernel each way
    instead of cast + multiply + add (and their autograd counterparts)."""

    @staticmethod
    def forward(ctx, x, y, gamma):
        lib = _lib.load()
        C = x.shape[-1]
        M = x.numel() // C
        gf = _f32(gamma) if gamma is not None else None
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_scale_residual(x.data_ptr(), _code(x), y.data_ptr(), _lib.ptr(gf), out.data_ptr(), _code(out), M, C,
                                          _stream()), "cnx_scale_residual")
        ctx.save_for_backward(y if (gamma is not None and ctx.needs_input_grad[2]) else None, gf)
        ctx.x_dtype, ctx.C, ctx.M = x.dtype, C, M
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, gf = ctx.saved_tensors
        C, M = ctx.C, ctx.M
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        want_p = gf is not None and ctx.needs_input_grad[2] and not _INPUT_GRAD_ONLY
        dy = torch.empty(g.shape, device=g.device, dtype=torch.bfloat16) if ctx.needs_input_grad[1] else None
        dgamma = db = ws = None
        if want_p:
            dgamma = torch.empty(C, device=g.device, dtype=torch.float32)
            db = torch.empty(C, device=g.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_colsum_ws_floats(C), device=g.device, dtype=torch.float32)
        if dy is not None or want_p:
            _lib.check(lib.cnx_scale_residual_bwd(g.data_ptr(), _code(g), _lib.ptr(y) if want_p else None, _lib.ptr(gf), _lib.ptr(dy),
                                                  _lib.ptr(dgamma), _lib.ptr(db), _lib.ptr(ws), M, C, _stream()),
                       "cnx_scale_residual_bwd")
        dx = g if ctx.needs_input_grad[0] else None
        if dx is not None and dx.dtype != ctx.x_dtype:
            dx = dx.to(ctx.x_dtype)
        return dx, dy, dgamma


def scale_residual(x, y, gamma=None):
    """``x + gamma * y`` (fp32 result) for a bf16 branch output ``y``; eager composition off the device / for other dtypes."""
    if (MODE == "eager" or not x.is_cuda or y.dtype != torch.bfloat16 or x.dtype not in (torch.float32, torch.bfloat16)
            or x.shape != y.shape or x.shape[-1] % 4 != 0):
        return x + (y * gamma if gamma is not None else y)
    return _ScaleResidual.apply(x.contiguous(), y.contiguous(), gamma)


def layer_norm_last(x, weight, bias, eps):
    """LayerNorm over the last dim of ``[..., C]`` (the ViT blocks' ``nn.LayerNorm``, timm ``vision_transformer.Block``): one
    kernel, output already in the autocast activation dtype (the eager pair is an fp32 LayerNorm + a cast in the next Linear)."""
    if MODE == "eager" or not x.is_cuda or x.shape[-1] % 4 != 0 or not _hip_dtype_ok(x):
        return F.layer_norm(x, weight.shape, weight, bias, eps)
    return _ln_rows(x.contiguous(), weight, bias, eps, False)


def layer_norm_cf(x, weight, bias, eps):
    """LayerNorm over dim 1 of ``[N,C,H,W]`` (``utils_architecture.py:76-81``; timm LayerNorm2d)."""
    if MODE == "eager" or (x.is_cuda and not _hip_dtype_ok(x)):
        return F.layer_norm(x.permute(0, 2, 3, 1), weight.shape, weight, bias, eps).permute(0, 3, 1, 2)
    return _ln_rows(_rows(x), weight, bias, eps, False).permute(0, 3, 1, 2)


def layer_norm_cf_gelu(x, weight, bias, eps):
    """``GELU(LN_cf(x))`` — the ConvStem pair (``utils_architecture.py:128-129`` etc.), one kernel."""
    if MODE == "eager" or (x.is_cuda and not _hip_dtype_ok(x)):
        return F.gelu(F.layer_norm(x.permute(0, 2, 3, 1), weight.shape, weight, bias, eps).permute(0, 3, 1, 2))
    return _ln_rows(_rows(x), weight, bias, eps, True).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ stage downsample: LN + 2x2/2 convolution
class _DownsampleLnConv(torch.autograd.Function):
    """``Conv2d(C, C', 2, stride 2)(LayerNorm2d(x))`` (``models/convnext.py:76-83``) on channels-last rows:
    ``cnx_layernorm_fwd_patch2`` writes LN(x) directly in 2x2-patch form ``[N*H/2*W/2, 4C]``, the convolution is one library
    GEMM with the bias in its epilogue (instead of an implicit-GEMM convolution + a separate bias add), its input gradient a
    GEMM whose result ``cnx_layernorm_bwd_patch2`` reads in the same form; filter gradient = split-K batched GEMM."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, weight, bias):
        lib = _lib.load()
        N, H, W, C = x.shape
        Co = weight.shape[0]
        Mo = N * (H // 2) * (W // 2)
        yp = torch.empty(Mo, 4 * C, device=x.device, dtype=torch.bfloat16)
        mean = torch.empty(N * H * W, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lw, lb = _f32(ln_w), _f32(ln_b)
        _lib.check(lib.cnx_layernorm_fwd_patch2(x.data_ptr(), _code(x), lw.data_ptr(), lb.data_ptr(), eps, yp.data_ptr(), _code(yp),
                                                mean.data_ptr(), rstd.data_ptr(), N, H, W, C, _stream()), "cnx_layernorm_fwd_patch2")
        wp = _cached((weight,), "patch2_bf16", lambda w: w.permute(0, 2, 3, 1).reshape(Co, 4 * C).to(torch.bfloat16).contiguous())
        if _gemm_ok(yp, wp, "downsample"):
            out = _gemm_nt(yp, wp, EPI_BIAS, bias=_f32(bias) if bias is not None else None)
        elif bias is not None:
            bb = _cached((bias,), "bf16", lambda b: b.to(torch.bfloat16).contiguous())
            out = torch.addmm(bb, yp, wp.t())
        else:
            out = yp @ wp.t()
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, lw, mean, rstd, yp, wp)
            ctx.has_bias, ctx.weight = bias is not None, weight
        return out.view(N, H // 2, W // 2, Co)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, lw, mean, rstd, yp, wp = ctx.saved_tensors
        N, H, W, C = x.shape
        Co = wp.shape[0]
        nig = ctx.needs_input_grad
        want_p = any(nig[1:]) and not _INPUT_GRAD_ONLY
        g2 = g.reshape(-1, Co)
        gb = g2 if g2.dtype == torch.bfloat16 else g2.to(torch.bfloat16)
        gb = gb.contiguous()
        dx = dlw = dlb = dwt = db = None
        if _gemm_dims_ok(gb, Co, 4 * C, "downsample"):                           # [Mo, 4C] in patch form
            wpt = _cached((ctx.weight,), "patch2_bf16_t",
                          lambda w: w.permute(0, 2, 3, 1).reshape(Co, 4 * C).to(torch.bfloat16).t().contiguous())
            dyp = _gemm_nt(gb, wpt, EPI_BIAS)
        else:
            dyp = gb @ wp
        dx = torch.empty_like(x)
        ws = None
        if want_p:
            dlw = torch.empty(C, device=x.device, dtype=torch.float32)
            dlb = torch.empty(C, device=x.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_layernorm_bwd_patch2(dyp.data_ptr(), _code(dyp), x.data_ptr(), _code(x), lw.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), dx.data_ptr(), _code(dx), _lib.ptr(dlw), _lib.ptr(dlb), _lib.ptr(ws),
                                                N, H, W, C, _stream()), "cnx_layernorm_bwd_patch2")
        if want_p:
            dwt = _wgrad(gb, yp).view(Co, 2, 2, C).permute(0, 3, 1, 2)          # [Co, C, 2, 2] like the parameter
            if ctx.has_bias:
                if g2.dtype == torch.bfloat16 and g2.is_contiguous() and Co % 4 == 0:
                    # the column sums by the one-pass kernel (fixed-order partials) instead of torch's reduction (1.3 TB/s on [200704, 192]).
                    # bf16 gradients only (the autocast backward): the kernel sums its bf16-ROUNDED dO = g * 1, which is g itself there
                    # and would not be the fp32 column sum of an fp32 gradient
                    db = torch.empty(Co, device=g2.device, dtype=torch.float32)
                    unused_dgamma = torch.empty(Co, device=g2.device, dtype=torch.float32)      # the kernel's d(gamma) output: no y, no meaning
                    ws2 = torch.empty(lib.cnx_colsum_ws_floats(Co), device=g2.device, dtype=torch.float32)
                    _lib.check(lib.cnx_scale_residual_bwd(g2.data_ptr(), _code(g2), None, None, None, unused_dgamma.data_ptr(), db.data_ptr(),
                                                          ws2.data_ptr(), g2.shape[0], Co, _stream()), "cnx_scale_residual_bwd(colsum)")
                else:
                    db = g2.sum(0, dtype=torch.float32)
        return dx, dlw, dlb, None, dwt, db


def downsample_supported(x, ln_w, conv):
    """LayerNorm2d + Conv2d(kernel 2, stride 2, no padding / dilation / groups) on a CUDA tensor under bf16 activations with
    even H, W and a width the wide LayerNorm kernels take."""
    if MODE == "eager" or not x.is_cuda or x.dim() != 4:
        return False
    C = x.shape[1]
    ok_conv = (conv.kernel_size == (2, 2) and conv.stride == (2, 2) and conv.padding == (0, 0) and conv.dilation == (1, 1)
               and conv.groups == 1 and conv.in_channels == C)
    g = C // 24
    ok_c = C % 24 == 0 and 2 <= g <= 32 and (g & (g - 1)) == 0
    return bool(ok_conv and ok_c and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and _act_dtype(x) == torch.bfloat16
                and x.dtype in (torch.float32, torch.bfloat16))


def downsample_ln_conv(x, ln_w, ln_b, eps, weight, bias):
    """[N,C,H,W] -> [N,C',H/2,W/2] (NCHW-shaped view of channels-last rows)."""
    return _DownsampleLnConv.apply(_rows(x), ln_w, ln_b, float(eps), weight, bias).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ first ConvStem convolution
# True: the ConvStem convolutions' filter / bias gradients on csrc/wgrad_kernels.hip (round 5); False: the library's
# convolution_backward (MIOpen) - kept for the A/B and the parity tests
STEM_WGRAD_HIP = os.environ.get("APGD_STEM_WGRAD", "hip") != "lib"


def _stem_wgrad(lib, x, gr, weight, has_bias):
    """Filter / bias gradient of the first ConvStem convolution from the fp32 NCHW image ``x`` and the bf16 NHWC rows ``gr`` of the
    output gradient (``utils_architecture.py:205-211`` backward)."""
    N, _, H, W = x.shape
    P = weight.shape[0]
    if STEM_WGRAD_HIP:
        dw = torch.empty(weight.shape, device=x.device, dtype=torch.float32)
        db = torch.empty(P, device=x.device, dtype=torch.float32) if has_bias else None
        ws = torch.empty(lib.cnx_stem_conv_wgrad_ws_floats(P), device=x.device, dtype=torch.float32)
        _lib.check(lib.cnx_stem_conv_wgrad(x.data_ptr(), gr.data_ptr(), dw.data_ptr(), _lib.ptr(db), ws.data_ptr(), N, H, W, P,
                                           _stream()), "cnx_stem_conv_wgrad")
        return dw.to(weight.dtype), db
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gb = gr.permute(0, 3, 1, 2)                                    # NCHW view of channels-last memory
    # (filter gradient from the library, the bias gradient never: conv_bias_grad)
    _, dw, _ = torch.ops.aten.convolution_backward(gb, xb, weight.to(torch.bfloat16), None, [2, 2], [1, 1], [1, 1],
                                                   False, [0, 0], 1, [False, True, False])
    return dw.to(weight.dtype), (conv_bias_grad(gb) if has_bias else None)


class _StemConv(torch.autograd.Function):
    """``Conv2d(3, P, 3, stride 2, padding 1)`` on the fp32 NCHW image batch -> NCHW-shaped view of NHWC bf16 rows.

    The layer that touches the attack state: its input gradient is what the APGD update consumes.  Forward and input
    gradient are ``cnx_stem_conv_fwd`` / ``cnx_stem_conv_dgrad`` (no cast, no layout copy, bias fused); the filter / bias
    gradient (training backward only) stays in the library."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        N, _, H, W = x.shape
        P = weight.shape[0]
        wq = _cached((weight,), "stem_wq", _pack_stem)
        bf = _f32(bias) if bias is not None else None
        out = torch.empty(N, (H + 1) // 2, (W + 1) // 2, P, device=x.device, dtype=torch.bfloat16)
        _lib.check(lib.cnx_stem_conv_fwd(x.data_ptr(), wq.data_ptr(), _lib.ptr(bf), out.data_ptr(), N, H, W, P, _stream()),
                   "cnx_stem_conv_fwd")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, weight, wq)
            ctx.has_bias = bias is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight, wq = ctx.saved_tensors
        N, _, H, W = x.shape
        P = weight.shape[0]
        gr = g.permute(0, 2, 3, 1)
        if gr.dtype != torch.bfloat16 or not gr.is_contiguous():
            gr = gr.to(torch.bfloat16).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _stem_dgrad(lib, x, gr, wq, N, H, W, P)
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY:
            dw, db = _stem_wgrad(lib, x, gr, weight, ctx.has_bias)
        return dx, dw, db


class _StemConvLnGelu(torch.autograd.Function):
    """First ConvStem convolution + LayerNorm(channels) + GELU in ONE forward kernel (``cnx_stem_conv_ln_gelu_fwd``): the
    [N,112,112,P] convolution output is normalised while it sits in LDS instead of being written, re-read and re-written by a
    separate LayerNorm pass; a gradient-free forward does not write it at all.  Backward: ``cnx_layernorm_bwd(gelu=1)`` on the
    saved convolution output, then ``_StemConv``'s input / filter gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, ln_w, ln_b, eps):
        lib = _lib.load()
        N, _, H, W = x.shape
        P = weight.shape[0]
        wq = _cached((weight,), "stem_wq", _pack_stem)
        bf = _f32(bias) if bias is not None else None
        lw, lb = _f32(ln_w), _f32(ln_b)
        need_grad = any(ctx.needs_input_grad)
        OH, OW = (H + 1) // 2, (W + 1) // 2
        act = torch.empty(N, OH, OW, P, device=x.device, dtype=torch.bfloat16)
        y = mean = rstd = None
        if need_grad:
            y = torch.empty(N, OH, OW, P, device=x.device, dtype=torch.bfloat16)
            mean = torch.empty(N * OH * OW, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
        _lib.check(lib.cnx_stem_conv_ln_gelu_fwd(x.data_ptr(), wq.data_ptr(), _lib.ptr(bf), lw.data_ptr(), lb.data_ptr(), eps,
                                                 _lib.ptr(y), act.data_ptr(), _lib.ptr(mean), _lib.ptr(rstd), N, H, W, P, _stream()),
                   "cnx_stem_conv_ln_gelu_fwd")
        if need_grad:
            ctx.save_for_backward(x, weight, wq, y, mean, rstd, lw, lb)
            ctx.has_bias = bias is not None
        return act.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight, wq, y, mean, rstd, lw, lb = ctx.saved_tensors
        N, _, H, W = x.shape
        P = weight.shape[0]
        M = y.numel() // P
        nig = ctx.needs_input_grad
        gr = g.permute(0, 2, 3, 1)
        if gr.dtype not in (torch.float32, torch.bfloat16) or not gr.is_contiguous():
            gr = gr.to(torch.bfloat16).contiguous()
        want_ln = (nig[3] or nig[4]) and not _INPUT_GRAD_ONLY
        dlw = dlb = ws = None
        if want_ln:
            dlw = torch.empty(P, device=x.device, dtype=torch.float32)
            dlb = torch.empty(P, device=x.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(P), device=x.device, dtype=torch.float32)
        dy = torch.empty_like(y)                                             # gradient w.r.t. the convolution output, bf16 NHWC
        _lib.check(lib.cnx_layernorm_bwd(gr.data_ptr(), _code(gr), y.data_ptr(), _code(y), lw.data_ptr(), lb.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), _code(dy), _lib.ptr(dlw), _lib.ptr(dlb),
                                         _lib.ptr(ws), M, P, 1, _stream()), "cnx_layernorm_bwd")
        dx = dw = db = None
        if nig[0]:
            dx = _stem_dgrad(lib, x, dy, wq, N, H, W, P)
        if (nig[1] or nig[2]) and not _INPUT_GRAD_ONLY:
            dw, db = _stem_wgrad(lib, x, dy, weight, ctx.has_bias)
        return dx, dw, db, dlw, dlb, None


_STEM_LN_FUSED = os.environ.get("APGD_STEM_LN_FUSED", "0") not in ("0", "")


def stem_fused_ln():
    """The one-kernel stem (convolution + LayerNorm + GELU on the tile in LDS) or convolution + LayerNorm-GELU kernel.  Measured on
    MI355X (``tools/stem_bench.py``, batch 256 / 128, 224 x 224, after the convolution's move to 16-byte quad loads - 112 us alone,
    162 before): one kernel 214 / 111 us for a gradient-free forward and 234 / 139 us with the saved convolution output, two kernels
    282 / 147 and 278 / 141 us.  In the step (three interleaved pairs, ``gpurun_out/r4sf``) the two compositions are inside each
    other's noise (49.16 / 49.49 / 49.32 vs 49.78 / 49.22 / 49.28 ms): the two-kernel composition stays the default, the fused kernel
    stays in the library behind this constant."""
    return _STEM_LN_FUSED


def stem_conv_ln_gelu(x, weight, bias, ln_w, ln_b, eps):
    return _StemConvLnGelu.apply(x, weight, bias, ln_w, ln_b, float(eps))


def _pack_stem(w):
    lib = _lib.load()
    P = w.shape[0]
    w = w.contiguous()
    if w.dtype not in (torch.float32, torch.bfloat16):
        w = w.float()
    wq = torch.empty(lib.cnx_stem_conv_packed_bytes(P), device=w.device, dtype=torch.uint8)
    _lib.check(lib.cnx_stem_conv_pack(w.data_ptr(), _code(w), wq.data_ptr(), P, _stream()), "cnx_stem_conv_pack")
    return wq


def stem_conv_supported(x, weight, stride, padding):
    return (MODE != "eager" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and x.shape[1] == 3 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and tuple(weight.shape[1:]) == (3, 3, 3)
            and x.numel() < 2 ** 30                               # 32-bit buffer offsets in the kernel (it returns APGD_ERR_ARG beyond)
            and tuple(stride) == (2, 2) and tuple(padding) == (1, 1) and torch.is_autocast_enabled()
            and torch.get_autocast_dtype('cuda') == torch.bfloat16 and bool(_lib.load().cnx_stem_conv_supported(weight.shape[0])))


def stem_conv(x, weight, bias):
    return _StemConv.apply(x, weight, bias)


# ------------------------------------------------------------------------------ second ConvStem convolution
def _pack_conv2(w):
    lib = _lib.load()
    CO, CI = w.shape[0], w.shape[1]
    w = w.contiguous()
    if w.dtype not in (torch.float32, torch.bfloat16):
        w = w.float()
    pk = torch.empty(lib.cnx_conv3x3s2_packed_elems(CI, CO), device=w.device, dtype=torch.bfloat16)
    _lib.check(lib.cnx_conv3x3s2_pack(w.data_ptr(), _code(w), pk.data_ptr(), CI, CO, _stream()), "cnx_conv3x3s2_pack")
    return pk


# 2: forward and input gradient through cnx_conv3x3s2_fwd / _dgrad (1: forward only; 0: library - measured behind, kept for A/B by attribute)
_CONV2_MODE = 2


def conv3x3s2_supported(x, conv):
    """3x3 / stride 2 / padding 1 convolution of a channels-last bf16 activation with a kernel for its widths and map size."""
    if MODE == "eager" or not _CONV2_MODE or not x.is_cuda or x.dim() != 4 or x.dtype != torch.bfloat16:
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (2, 2) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1:
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    return bool(_lib.load().cnx_conv3x3s2_supported(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3]))


class _Conv3x3s2(torch.autograd.Function):
    """``Conv2d(CI, CO, 3, stride 2, padding 1)`` on a channels-last bf16 activation (NCHW-shaped views of NHWC rows in and out).
    Forward = ``cnx_conv3x3s2_fwd`` (implicit GEMM on MFMA, filter resident in LDS); the input gradient is the library's
    the input gradient ``cnx_conv3x3s2_dgrad`` (2 x 2 input patches against their 2 x 2 output neighbours); filter / bias gradients stay in
    the library."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        N, CI, H, W = x.shape
        CO = weight.shape[0]
        pk = _cached((weight,), "conv2_packed", _pack_conv2)
        bf = _f32(bias) if bias is not None else None
        xr = x.permute(0, 2, 3, 1)
        out = torch.empty(N, H // 2, W // 2, CO, device=x.device, dtype=torch.bfloat16)
        _lib.check(lib.cnx_conv3x3s2_fwd(xr.data_ptr(), pk.data_ptr(), _lib.ptr(bf), out.data_ptr(), N, H, W, CI, CO, _stream()),
                   "cnx_conv3x3s2_fwd")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, weight, pk)
            ctx.has_bias = bias is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight, pk = ctx.saved_tensors
        N, CI, H, W = x.shape
        CO = weight.shape[0]
        if g.dtype != torch.bfloat16 or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        want_w = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY
        dx = dw = db = None
        if ctx.needs_input_grad[0] and _CONV2_MODE >= 2:
            dxr = torch.empty(N, H, W, CI, device=x.device, dtype=torch.bfloat16)
            _lib.check(lib.cnx_conv3x3s2_dgrad(g.permute(0, 2, 3, 1).data_ptr(), pk.data_ptr(), dxr.data_ptr(), N, H, W, CI, CO,
                                               _stream()), "cnx_conv3x3s2_dgrad")
            dx = dxr.permute(0, 3, 1, 2)
        need_lib_dx = ctx.needs_input_grad[0] and dx is None
        if want_w and STEM_WGRAD_HIP and lib.cnx_conv3x3s2_wgrad_supported(CI, CO, H, W):
            # filter / bias gradient on csrc/wgrad_kernels.hip, in the weight's channels-last order
            dwf = torch.empty(CO, 3, 3, CI, device=x.device, dtype=torch.float32)
            db = torch.empty(CO, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            ws = torch.empty(lib.cnx_conv3x3s2_wgrad_ws_floats(CI, CO), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_conv3x3s2_wgrad(x.permute(0, 2, 3, 1).data_ptr(), g.permute(0, 2, 3, 1).data_ptr(), dwf.data_ptr(),
                                               _lib.ptr(db), ws.data_ptr(), N, H, W, CI, CO, _stream()), "cnx_conv3x3s2_wgrad")
            dw = dwf.permute(0, 3, 1, 2).to(weight.dtype)
            if db is not None:
                db = db.to(weight.dtype)
            want_w = False
        if need_lib_dx or want_w:
            wl = _cached((weight,), "bf16_cl", lambda w: w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
            # (never the library's bias gradient: under hipGraph replay MIOpen's returned non-finite values, see conv_bias_grad)
            r = torch.ops.aten.convolution_backward(g, x, wl, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                    [bool(need_lib_dx), bool(want_w), False])
            if need_lib_dx:
                dx = r[0]
            if want_w:
                dw = r[1].to(weight.dtype)
                db = conv_bias_grad(g).to(weight.dtype) if ctx.has_bias else None
        return dx, dw, db


def conv3x3s2(x, weight, bias):
    return _Conv3x3s2.apply(x, weight, bias)


def conv_bias_grad(g):
    """``d(loss)/d(bias)`` of a convolution = the sum of the output gradient ``g`` [N, C, H, W] over N, H, W, in fp32, by a plain
    reduction of ours / ATen's - never the library's.  Found in round 6 (``tools/ab_nan_hunt.py``, ``gpurun_out/r6z``): under hipGraph
    REPLAY of the training pass MIOpen's bias gradient of a ConvStem convolution came back non-finite (ConvNeXt-B, the 96 -> 128
    stride-1 convolution: 5 of 6 fresh trainers within ten steps; the 48 -> 96 stride-2 one through the library: 1 of 5) while the
    eager pass never showed it - one NaN bias gradient, and AdamW has put a NaN into the model."""
    if g.is_cuda and g.dim() == 4 and g.dtype == torch.bfloat16 and g.is_contiguous(memory_format=torch.channels_last) and MODE != "eager" \
            and g.shape[1] % 4 == 0 and g.numel() > 0:
        lib = _lib.load()
        N, C, H, W = g.shape
        rows = g.permute(0, 2, 3, 1).reshape(N * H * W, C)              # a view: channels-last rows
        db = torch.empty(C, device=g.device, dtype=torch.float32)
        unused_dgamma = torch.empty(C, device=g.device, dtype=torch.float32)
        ws = torch.empty(lib.cnx_colsum_ws_floats(C), device=g.device, dtype=torch.float32)
        _lib.check(lib.cnx_scale_residual_bwd(rows.data_ptr(), _code(rows), None, None, None, unused_dgamma.data_ptr(), db.data_ptr(),
                                              ws.data_ptr(), rows.shape[0], C, _stream()), "cnx_scale_residual_bwd(colsum)")
        return db
    return g.sum((0, 2, 3), dtype=torch.float32)


class _ConvLib(torch.autograd.Function):
    """A convolution that stays with the library (the ConvStem convolutions without a hand-written kernel: ``ConvBlock3``'s second and
    third, the ViT stems' later ones, the 1x1 projections) with ITS backward asked for the input and filter gradients only; the bias
    gradient is ``conv_bias_grad``.  The casts autocast would make are made here (``aten.convolution`` on already-cast operands)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled()) else None
        xc = x if dt is None or x.dtype == dt else x.to(dt)
        if dt is None:
            wc = weight
        else:
            wc = _cached((weight,), "conv_cast", lambda w: w.to(dt)) if weight.dtype != dt else weight
        bc = None if bias is None else (bias if dt is None or bias.dtype == dt else bias.to(dt))
        out = torch.ops.aten.convolution(xc, wc, bc, list(stride), list(padding), list(dilation), False, [0, 0], groups)
        ctx.save_for_backward(xc, wc)
        ctx.conf = (list(stride), list(padding), list(dilation), groups, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xc, wc = ctx.saved_tensors
        stride, padding, dilation, groups, x_dt, w_dt, b_dt = ctx.conf
        nig = ctx.needs_input_grad
        want_w = nig[1] and not _INPUT_GRAD_ONLY
        want_b = b_dt is not None and nig[2] and not _INPUT_GRAD_ONLY
        if g.dtype != xc.dtype:
            g = g.to(xc.dtype)
        dx = dw = db = None
        if nig[0] or want_w:
            r = torch.ops.aten.convolution_backward(g, xc, wc, None, stride, padding, dilation, False, [0, 0], groups,
                                                    [bool(nig[0]), bool(want_w), False])
            dx = r[0].to(x_dt) if nig[0] else None
            dw = r[1].to(w_dt) if want_w else None
        if want_b:
            db = conv_bias_grad(g).to(b_dt)
        return dx, dw, db, None, None, None, None


def conv2d_lib(x, conv):
    """``conv(x)`` for an ``nn.Conv2d`` that has no hand-written kernel, with the bias gradient kept away from the library."""
    if MODE == "eager" or not x.is_cuda or conv.bias is None or conv.padding_mode != "zeros" or isinstance(conv.padding, str):
        return F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
    return _ConvLib.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)


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


# ------------------------------------------------------------------------------ fused block tail (MFMA)
_wcache = {}


CACHE_STATS = {"miss": 0}
_WEIGHTS_EPOCH = 0


def invalidate_weight_cache(*_):
    """Drop every derived weight copy.  ``Tensor._version`` is NOT bumped by fused optimizers (``AdamW(fused=True)`` updates
    parameters in place without touching the counter), so the cache is also keyed on this epoch, advanced by a global
    optimizer-step hook; call it by hand after any other out-of-band weight update."""
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(invalidate_weight_cache)
except ImportError:                                      # older torch: ATTrainStep calls invalidate_weight_cache() itself
    pass


def load_gemm_table(path=None):
    """Point PyTorch's TunableOp at the shipped hipBLASLt / rocBLAS solution table for the GEMM shapes of the benchmark
    configuration and of BASELINE configs #3 / #4 (``gemm_tuning_gfx950.csv``, recorded on MI355X with
    ``PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=... python tools/probe/cfg_profile.py
    <arch> ...``, refreshed at the end of round 2: 58.5 -> 57.6 ms for the ConvNeXt-T step, 139.9 -> 136.3 ms ViT-B, 340.2 ->
    333.9 ms ConvNeXt-L @320, 30.3 -> 25.9 ms per iteration of the fp32 evaluation attack on ConvNeXt-B).  Lookup only: no tuning at run time, shapes that are not in the table (and library builds other
    than the ones in its ``Validator`` lines) use the default heuristic.  ``APGD_GEMM_TABLE=0`` disables it."""
    if os.environ.get("APGD_GEMM_TABLE", "1") == "0" or not torch.cuda.is_available():
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):          # the user drives TunableOp themselves
        return False
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning_gfx950.csv")
    if not os.path.exists(path):
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.record_untuned_enable(False)
        return bool(tunable.read_file(path))                 # read only: the shipped table is never written
    except Exception:                                        # an older / different PyTorch build: default GEMM selection
        return False


# The training-pass graph reads the derived copies of the attack program that replays in front of it instead of rebuilding them
# (train_step._TrainPassGraph, round 6): each copy is made once per step.
SHARE_DERIVED = os.environ.get("APGD_SHARE_DERIVED", "1") != "0"


# Set (to a dict) while graphed.py captures the attack into hipGraphs: derived weight copies are then built INSIDE the capture,
# once per capture, from the live parameters - a replay re-packs them on the device and never reads a copy made before the last
# optimizer step - and the ordinary cache is neither read nor written.
_CAPTURE_CACHE = None


# Number of the graph segment being captured (graphed._Recorder bumps it at every cut): a derived copy made in an EARLIER segment is
# complete by stream order when a later segment runs; one made in the SAME segment on another stream has to be waited for.
_CAPTURE_SEG = 0
_CHUNKED = False                                             # set by apgd._model_fwd_bwd_split while batch chunks run on several streams


def _await_producer(ev, st):
    """A derived copy is made on the stream that asks for it first.  When the attack's batch chunks run on two streams, the other
    chunk finds it in the cache moments later - its stream has to wait for the kernels that are still writing it (without this
    the second chunk could read the packed weights of the PREVIOUS optimizer step, or a mix)."""
    cur = torch.cuda.current_stream()
    if st is not None and cur != st:
        cur.wait_event(ev)


def _cached(params, tag, fn):
    """Derived copies of parameters (bf16 casts, MFMA-fragment packing), rebuilt only when a parameter changes
    (optimizer steps bump ``_version``); the attack's forwards and the train forward share them.  Entries hold weak
    references: ``id()`` and even the storage address of a dead parameter can be handed to a new one."""
    if _CAPTURE_CACHE is not None:
        ckey = tuple(id(q) for q in params) + (tag,)
        hit = _CAPTURE_CACHE.get(ckey)
        if hit is None:
            with torch.no_grad():
                val = fn(*[q.detach() for q in params])
            ev = st = None
            if _CHUNKED:
                st = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(st)
            hit = _CAPTURE_CACHE[ckey] = (val, params, ev, st, _CAPTURE_SEG)                 # params kept alive: ids stay unique
        elif _CHUNKED and hit[4] == _CAPTURE_SEG:
            _await_producer(hit[2], hit[3])
        return hit[0]
    key = tuple(id(q) for q in params) + (tag,)
    ver = (_WEIGHTS_EPOCH,) + tuple((q._version, q.data_ptr()) for q in params)
    hit = _wcache.get(key)
    if hit is not None and hit[0] == ver and all(r() is q for r, q in zip(hit[2], params)):
        if _CHUNKED:
            _await_producer(hit[3], hit[4])
        return hit[1]
    CACHE_STATS["miss"] += 1
    with torch.no_grad():
        val = fn(*[q.detach() for q in params])
    if len(_wcache) > 4096:                      # dead entries of models that no longer exist
        for k in [k for k, v in _wcache.items() if any(r() is None for r in v[2])]:
            del _wcache[k]
    ev = st = None
    if _CHUNKED and val.is_cuda:
        st = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(st)
    _wcache[key] = (ver, val, tuple(weakref.ref(q) for q in params), ev, st)
    return val


def _pack_mlp(w1, w2):
    """fc1 / fc2 weights -> bf16 in the order the fused kernel's MFMA lanes read them (``cnx_mlp_pack_weights``)."""
    lib = _lib.load()
    C = w1.shape[1]
    w1, w2 = w1.contiguous(), w2.contiguous()
    if w1.dtype != w2.dtype or w1.dtype not in (torch.float32, torch.bfloat16):
        w1, w2 = w1.float(), w2.float()
    wf = torch.empty(lib.cnx_mlp_packed_elems(C), device=w1.device, dtype=torch.bfloat16)
    _lib.check(lib.cnx_mlp_pack_weights(w1.data_ptr(), w2.data_ptr(), _code(w1), wf.data_ptr(), C, _stream()),
               "cnx_mlp_pack_weights")
    return wf


def _sum_parts(part):
    """fp32 sum over the leading (split-K) dimension of the bf16 partial products, one pass (``cnx_sum_parts_bf16``)."""
    S, L = part.shape[0], part[0].numel()
    if part.dtype != torch.bfloat16 or not part.is_cuda or L % 8 != 0 or not part.is_contiguous():
        return part.sum(0, dtype=torch.float32)
    out = torch.empty(part.shape[1:], device=part.device, dtype=torch.float32)
    _lib.check(_lib.load().cnx_sum_parts_bf16(part.data_ptr(), out.data_ptr(), S, L, _stream()), "cnx_sum_parts_bf16")
    return out


# ------------------------------------------------------------------------------ hand-written GEMM with fused epilogues
# APGD_GEMM selects where cnx_gemm_nt (csrc/gemm_kernels.hip: bias / GELU / layer scale + residual / GELU' in the epilogue) replaces
# the hipBLASLt GEMM + one-pass kernels of round 2:
#   "hip"  : every forward / input-gradient GEMM of the library-path ConvNeXt blocks, the downsample layers and the ViT linears;
#   "auto" : (default) only where it measured at least as fast in the step on MI355X - the stage downsample layers (K = 4C <= 1536,
#            N <= 768: 49 - 77 us against 55 - 90 us).  At the MLP shapes the kernel reaches 480 - 860 TFLOP/s against the library's
#            530 - 1240: it is bound by what a CU ingests from L2 (15 B / cycle / CU measured, 110 FLOP per staged byte at 256 x 192
#            tiles; profiles/r03_gemm.md), and the AT step was 3 ms slower with it everywhere;
#   "lib"  : nowhere (A/B timing).
# Weight gradients (contraction over M) are split-K library GEMMs in every mode.
_GEMM_MODE = os.environ.get("APGD_GEMM", "auto")
EPI_BIAS, EPI_BIAS_GELU, EPI_SCALE_RES, EPI_GELU_GRAD = 0, 1, 2, 3


# auto policy, "mlp" / "linear" sites of the TRAINING pass: cnx_gemm_nt up to this many (rows x wider dimension); 0 = never
# (default).  Kernel for kernel it equals the library on the ConvNeXt-T / -S layers (50176 x 1536 at C = 384, 12544 x 3072 at
# C = 768: +0.6 ms per step in the kernel trace, hipBLASLt 12 % -> 6 % of the step) but the un-profiled step was 2 ms slower with
# it in three interleaved runs (55.3 / 53.3 / 55.5 ms, 8e7 / 0 / 8e7), and on the larger problems (ViT-B 50432 x 3072, ConvNeXt-L)
# the library's kernels are 15 - 20 % ahead (profiles/r03_gemm.md) - so the training pass keeps the library
_GEMM_AUTO_MAX = int(float(os.environ.get("APGD_GEMM_AUTO_MAX", "0")))


def _gemm_on(site, M=0, N=0, K=0):
    """Is cnx_gemm_nt enabled for this call site ("mlp", "linear", "downsample", "direct") and problem size?"""
    if MODE == "eager" or _GEMM_MODE == "lib":
        return False
    if _GEMM_MODE == "hip" or site in ("downsample", "direct") or _ATTACK_PASS:
        return True
    return 0 < M * max(N, K) <= _GEMM_AUTO_MAX


def _gemm_ok(a, w_nk, site="direct"):
    """cnx_gemm_nt takes ``a`` [M, K] and ``w_nk`` [N, K]: bf16, unit inner stride, 16-byte rows, K % 64 == 0, N % 4 == 0."""
    return (a.dim() == 2 and w_nk.dim() == 2 and _gemm_on(site, a.shape[0], w_nk.shape[0], a.shape[1])
            and a.is_cuda and a.dtype == torch.bfloat16 and w_nk.dtype == torch.bfloat16
            and a.stride(1) == 1 and w_nk.stride(1) == 1 and a.shape[1] == w_nk.shape[1]
            and a.shape[1] % 64 == 0 and w_nk.shape[0] % 4 == 0 and a.stride(0) % 8 == 0 and w_nk.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and w_nk.data_ptr() % 16 == 0 and a.shape[0] > 0)


def _gemm_dims_ok(a, K, N, site="direct"):
    """The same test for an operand pair that does not exist yet: ``a`` is a bf16 row matrix of ours, the weight copy will be
    a fresh contiguous [N, K] bf16 tensor."""
    return (_gemm_on(site, a.shape[0], N, K) and a.is_cuda and a.dtype == torch.bfloat16 and K % 64 == 0 and N % 4 == 0
            and a.shape[0] > 0)


def _gemm_nt(a, w_nk, epi=EPI_BIAS, bias=None, gamma=None, resid=None, out_dtype=torch.bfloat16, z_out=None, z_in=None):
    """``epilogue(a @ w_nk^T)`` through ``cnx_gemm_nt`` (see include/convnext_hip.h).  ``bias`` / ``gamma`` fp32 [N]."""
    lib = _lib.load()
    M, K = a.shape
    N = w_nk.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    z = z_out if z_out is not None else z_in
    _lib.check(lib.cnx_gemm_nt(a.data_ptr(), a.stride(0), w_nk.data_ptr(), w_nk.stride(0), out.data_ptr(), N, _code(out), M, N, K, epi,
                               _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(resid), N, _code(resid) if resid is not None else 0,
                               _lib.ptr(z_out), _lib.ptr(z_in), z.stride(0) if z is not None else N, _stream()), "cnx_gemm_nt")
    return out


def _bf16(w):
    return _cached((w,), "bf16", lambda t: t.to(torch.bfloat16).contiguous())


def _bf16_t(w):
    """[out, in] weight -> its transpose [in, out] in bf16: the K-contiguous B operand of an input-gradient GEMM."""
    return _cached((w,), "bf16_t", lambda t: t.to(torch.bfloat16).t().contiguous())


# split-K factor of the weight-gradient GEMMs.  Measured on MI355X (tools/wgrad_bench.py, bmm + partial sum): the optimum puts
# about one 192 x 192 output tile on every CU, S * N1 * N2 ~ 256 * 192 * 192, over the whole range of shapes of the model
# (S = 4 for 768 x 3072 at M = 12 544 ... S = 256 for 384 x 96 at M = 802 816); two shapes sit one step below the rule.
_SPLIT_K_TARGET = 256 * 192 * 192
_SPLIT_K_MEASURED = {(768, 192, 200704): 32}


def _split_k(M, N1, N2):
    S = _SPLIT_K_MEASURED.get((N1, N2, M))
    if S is None:
        S = 1 << max(0, round(math.log2(max(1.0, _SPLIT_K_TARGET / float(N1 * N2)))))
    while S > 1 and (M % S != 0 or M // S < 256):
        S //= 2
    return S


# "hip": the weight gradients (contractions over the rows of two row-major operands) on cnx_gemm_tn (csrc/wgrad_kernels.hip, round 5)
# wherever its shape guards hold; "lib": the split-K batched library GEMM + partial sums of rounds 1 - 4 (A/B, parity tests)
_WGRAD_MODE = os.environ.get("APGD_WGRAD", "hip")


def _wgrad(x, y):
    """``x^T y`` for tall operands (``x`` [M, N1], ``y`` [M, N2], M >> N) -> fp32 [N1, N2]: ``cnx_gemm_tn`` (both operands as
    they lie in memory, transposed on their way out of LDS), or the library composition ``_wgrad_lib``."""
    M, N1 = x.shape
    N2 = y.shape[1]
    if (_WGRAD_MODE == "hip" and MODE != "eager" and x.is_cuda and x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
            and x.stride(1) == 1 and y.stride(1) == 1 and x.stride(0) % 8 == 0 and y.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0 and M * max(x.stride(0), y.stride(0)) * 2 < 2 ** 32):
        lib = _lib.load()
        if lib.cnx_gemm_tn_supported(M, N1, N2):
            d = torch.empty(N1, N2, device=x.device, dtype=torch.float32)
            ws = torch.empty(max(4, lib.cnx_gemm_tn_ws_floats(M, N1, N2)), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_gemm_tn(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), d.data_ptr(), ws.data_ptr(), M, N1, N2,
                                       _stream()), "cnx_gemm_tn")
            return d
    return _wgrad_lib(x, y)


def _tn_ok(M, N1, N2):
    return _WGRAD_MODE == "hip" and MODE != "eager" and M * 4 * max(N1, N2) < 2 ** 32 and bool(_lib.load().cnx_gemm_tn_supported(M, N1, N2))


# d(gamma) of a fused block from dW2 / d(b2) (cnx_block_dgamma) instead of a pass over g and y2; APGD_DGAMMA=pass restores the pass
_DGAMMA_FROM_DW2 = os.environ.get("APGD_DGAMMA", "dw2") != "pass"
# the LayerNorm parameter gradients of a fused block from dW1 / d(b1) (cnx_block_dln); APGD_DLN=pass: the LayerNorm backward's own sums
_DLN_FROM_DW1 = os.environ.get("APGD_DLN", "dw1") != "pass"
# ... and with them the LayerNorm backward itself in the epilogue of the block's training backward kernel (APGD_DLN=kernel: identity
# for the parameter gradients, but the separate LayerNorm backward kernel for du)
_LN_IN_TRAIN_BWD = os.environ.get("APGD_DLN", "dw1") == "dw1"


def _block_dln(lib, w1, dw1, db1, lw, lb, da, dhp_tiles, u, mean, rstd, M, C):
    dlw = torch.empty(C, device=u.device, dtype=torch.float32)
    dlb = torch.empty(C, device=u.device, dtype=torch.float32)
    # (workspace of the direct sums that ill-conditioned channels - |ln_b| > 4 |ln_w| - take instead of the identity)
    ws = torch.empty(lib.cnx_block_dln_ws_floats(C), device=u.device, dtype=torch.float32)
    _keep(ws)
    _lib.check(lib.cnx_block_dln(w1.data_ptr(), dw1.data_ptr(), db1.data_ptr(), lw.data_ptr(), lb.data_ptr(), _lib.ptr(da),
                                 dhp_tiles.data_ptr() if da is None else None, u.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 dlw.data_ptr(), dlb.data_ptr(), ws.data_ptr(), M, C, 4 * C, _stream()), "cnx_block_dln")
    return dlw, dlb


def _dgamma_needs_y2(M, C, w2):
    """Does the training backward of a fused block need the forward's pre-gamma output?  Not when d(gamma) comes from dW2 (the weight
    gradients on ``cnx_gemm_tn_ex`` over the kernels' tiles, from which the one channel in 2^24 with gamma == 0 is also served)."""
    return not (_DGAMMA_FROM_DW2 and w2.dtype == torch.float32 and w2.is_contiguous() and M % 32 == 0
                and _tn_ok(M, 4 * C, C) and _tn_ok(M, C, 4 * C))


def _block_dgamma(lib, g2, y2, h_tiles, gf, w2, b2f, dw2, db2, M, C):
    """``d(gamma)[c] = sum_m g*y2`` of ``models/convnext.py:47``.  From the second linear layer's weight gradients (one launch over
    [C, 4C]) when they were computed from ``dO = bf16(g*gamma)``; otherwise the one-pass sums of ``cnx_scale_residual_bwd``."""
    dgamma = torch.empty(C, device=g2.device, dtype=torch.float32)
    if _DGAMMA_FROM_DW2 and w2.dtype == torch.float32 and w2.is_contiguous() and dw2.is_contiguous() and (y2 is not None or M % 32 == 0):
        _lib.check(lib.cnx_block_dgamma(w2.data_ptr(), dw2.data_ptr(), _lib.ptr(b2f), _lib.ptr(db2) if b2f is not None else None, gf.data_ptr(),
                                        g2.data_ptr(), _code(g2), y2.reshape(M, C).data_ptr() if y2 is not None else None,
                                        h_tiles.data_ptr() if y2 is None else None, dgamma.data_ptr(), M, C, dw2.shape[1], _stream()),
                   "cnx_block_dgamma")
        return dgamma
    if y2 is None:
        raise _lib.ApgdHipError("d(gamma) by the one-pass sums needs the forward's pre-gamma output")
    db2_ = torch.empty(C, device=g2.device, dtype=torch.float32)
    ws = torch.empty(lib.cnx_colsum_ws_floats(C), device=g2.device, dtype=torch.float32)
    _keep(ws, db2_)
    _lib.check(lib.cnx_scale_residual_bwd(g2.data_ptr(), _code(g2), y2.reshape(M, C).data_ptr(), gf.data_ptr(), None,
                                          dgamma.data_ptr(), db2_.data_ptr(), ws.data_ptr(), M, C, _stream()), "cnx_scale_residual_bwd")
    return dgamma


def _wgrad_acc(a_t, a_acc, b_t, b_acc, M, N1, N2):
    """``A^T B`` -> (fp32 [N1, N2], fp32 [N1] column sums of A) through ``cnx_gemm_tn_ex`` with ONE operand in the accumulator-order
    tiles the fused block kernels write (``CNX_TN_ACC``: ``a_acc`` / ``b_acc``), the other a contiguous [M, N] bf16 row matrix."""
    lib = _lib.load()
    d = torch.empty(N1, N2, device=a_t.device, dtype=torch.float32)
    cs = torch.empty(N1, device=a_t.device, dtype=torch.float32)
    ws = torch.empty(lib.cnx_gemm_tn_ws_floats(M, N1, N2), device=a_t.device, dtype=torch.float32)
    _keep(ws)
    _lib.check(lib.cnx_gemm_tn_ex(a_t.data_ptr(), N1, 1 if a_acc else 0, b_t.data_ptr(), N2, 1 if b_acc else 0, d.data_ptr(),
                                  cs.data_ptr(), ws.data_ptr(), M, N1, N2, _stream()), "cnx_gemm_tn_ex")
    return d, cs


# both weight gradients of a fused block in ONE cnx_gemm_tn_pair launch (round 6; half the partial results of two cnx_gemm_tn_ex calls)
_TN_PAIR = os.environ.get("APGD_TN_PAIR", "1") != "0"


def _wgrad_block(dhp, a_rows, h, dos, M, C):
    """The two weight gradients of one block from the tiles / rows its backward kernel left: ``(dW1 [4C, C], db1 [4C], dW2 [C, 4C],
    db2 [C])`` = ``(dHpre^T LN(u), sum dHpre, dO^T H, sum dO)``; ``dhp``, ``h``: ``CNX_TN_ACC`` tiles, ``a_rows``, ``dos``: [M, C] bf16."""
    lib = _lib.load()
    if _TN_PAIR and lib.cnx_gemm_tn_pair_supported(M, 4 * C, C):
        dev = a_rows.device
        dw1 = torch.empty(4 * C, C, device=dev, dtype=torch.float32)
        dw2 = torch.empty(C, 4 * C, device=dev, dtype=torch.float32)
        db = torch.empty(5 * C, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.cnx_gemm_tn_pair_ws_floats(M, 4 * C, C), device=dev, dtype=torch.float32)
        _keep(ws)
        _lib.check(lib.cnx_gemm_tn_pair(dhp.data_ptr(), a_rows.data_ptr(), C, h.data_ptr(), dos.data_ptr(), C, dw1.data_ptr(), db.data_ptr(),
                                        dw2.data_ptr(), db[4 * C:].data_ptr(), ws.data_ptr(), M, 4 * C, C, _stream()), "cnx_gemm_tn_pair")
        return dw1, db[:4 * C], dw2, db[4 * C:]
    dw2, db2 = _wgrad_acc(dos, False, h, True, M, C, 4 * C)              # dO^T H        [C, 4C]
    dw1, db1 = _wgrad_acc(dhp, True, a_rows, False, M, 4 * C, C)         # dHpre^T LN(u) [4C, C]
    return dw1, db1, dw2, db2


def _wgrad_lib(x, y):
    """``x^T y`` for tall operands (``x`` [M, N1], ``y`` [M, N2], M >> N) -> fp32 [N1, N2].

    The weight gradients contract over M = N*H*W (802 816 rows at 56x56, batch 256) into a small result: as one GEMM that
    is a handful of output tiles with an enormous K (45 TFLOP/s in hipBLASLt at 384x96).  Split K into S batches of a batched
    GEMM (every batch fills its own output tiles, S of them fill the chip) and sum the S bf16 partial products in fp32."""
    M = x.shape[0]
    S = _split_k(M, x.shape[1], y.shape[1])
    if S == 1:
        return (x.t() @ y).float()
    part = torch.bmm(x.view(S, M // S, x.shape[1]).transpose(1, 2), y.view(S, M // S, y.shape[1]))
    return _sum_parts(part)


def _wgrad_t(xt, y):
    """Same contraction with the left operand already transposed: ``xt`` [N1, M] (K-contiguous), ``y`` [M, N2]."""
    N1, M = xt.shape
    S = _split_k(M, N1, y.shape[1])
    if S == 1:
        return (xt @ y).float()
    part = torch.bmm(xt.view(N1, S, M // S).transpose(0, 1), y.view(S, M // S, y.shape[1]))
    return _sum_parts(part)


# d(b1) of the fused blocks as an extra column of the dW1 GEMM (False: a separate reduction over [4C, M])
_DB1_IN_GEMM = True
_ONES_COL = {}


def _ones_col(device):
    t = _ONES_COL.get(device)
    if t is None:
        t = torch.zeros(8, device=device, dtype=torch.bfloat16)
        t[0] = 1
        _ONES_COL[device] = t
    return t


def _gelu_bf16(x):
    """GELU of a bf16 device tensor through ``cnx_gelu_fwd`` (no autograd: for use inside the autograd functions)."""
    if MODE == "eager" or not x.is_cuda or x.dtype != torch.bfloat16 or not x.is_contiguous() or x.numel() % 8 != 0:
        return F.gelu(x)
    y = torch.empty_like(x)
    _lib.check(_lib.load().cnx_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "cnx_gelu_fwd")
    return y


def _mlp_input_grads(lib, dos, hpre, w1, w2, w1b, w2b, db1, ws, M, C, want_da):
    """``dHpre = (dO W2) * GELU'(Hpre)`` (+ ``d(b1)`` column sums when ``db1`` is given) and ``da = dHpre W1``: two GEMMs with the
    GELU' in the first one's epilogue (``cnx_gemm_nt``), or library GEMMs around ``cnx_gelu_bwd_colsum``."""
    if _gemm_dims_ok(dos, C, 4 * C, "mlp") and _gemm_dims_ok(hpre, 4 * C, C, "mlp") and dos.is_contiguous() and hpre.is_contiguous():
        dhpre = _gemm_nt(dos, _bf16_t(w2), EPI_GELU_GRAD, z_in=hpre)                 # B = W2^T [4C, C]
        if db1 is not None:                                  # training pass: d(b1) = column sums of dHpre (one read)
            zeros = torch.empty(4 * C, device=dos.device, dtype=torch.float32)
            _lib.check(lib.cnx_scale_residual_bwd(dhpre.data_ptr(), _code(dhpre), None, None, None, zeros.data_ptr(), db1.data_ptr(),
                                                  ws.data_ptr(), M, 4 * C, _stream()), "cnx_scale_residual_bwd(colsum)")
        da = _gemm_nt(dhpre, _bf16_t(w1), EPI_BIAS) if want_da else None             # B = W1^T [C, 4C]
        return dhpre, da
    dh = dos @ w2b                                                               # [M, 4C]
    dhpre = torch.empty_like(dh)
    _lib.check(lib.cnx_gelu_bwd_colsum(dh.data_ptr(), hpre.data_ptr(), dhpre.data_ptr(), _lib.ptr(db1), _lib.ptr(ws), M, 4 * C,
                                       _stream()), "cnx_gelu_bwd_colsum")
    del dh
    return dhpre, ((dhpre @ w1b) if want_da else None)


class _MlpResidual(torch.autograd.Function):
    """``xs + gamma * fc2(GELU(fc1(h)))`` for the second half of a transformer block (timm ``Block.forward``:
    ``x + ls2(mlp(norm2(x)))``, ``Mlp`` = Linear, GELU, Linear): the two linears are library GEMMs with the bias in their
    epilogue, everything between and after them is one pass each - ``cnx_gelu_fwd``, ``cnx_scale_residual``; backward
    ``cnx_scale_residual_bwd`` (dO = g * gamma with d(gamma) and d(b2) column sums), ``cnx_gelu_bwd_colsum`` (GELU' with d(b1)),
    split-K weight gradients - instead of autograd's chain of casts, ATen GELU kernels and per-bias reductions."""

    @staticmethod
    def forward(ctx, xs, h, w1, b1, w2, b2, gamma):
        lib = _lib.load()
        C = h.shape[-1]
        M = h.numel() // C
        h2 = h.reshape(M, C)
        w1b = _cached((w1,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        w2b = _cached((w2,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        b1b = _cached((b1,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        b2b = _cached((b2,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
        gf = _f32(gamma) if gamma is not None else None
        need_grad = any(ctx.needs_input_grad)
        if _gemm_ok(h2, w1b, "mlp") and _gemm_dims_ok(h2, 4 * C, C, "mlp"):
            # fc1 + bias + GELU and fc2 + bias + layer scale + residual: two kernels, nothing element-wise between or after them
            hpre = torch.empty(M, 4 * C, device=h.device, dtype=torch.bfloat16) if need_grad else None
            hg = _gemm_nt(h2, w1b, EPI_BIAS_GELU, bias=_f32(b1), z_out=hpre)
            y2 = torch.empty(M, C, device=h.device, dtype=torch.bfloat16) if (need_grad and gf is not None) else None
            out = _gemm_nt(hg, w2b, EPI_SCALE_RES, bias=_f32(b2), gamma=gf, resid=xs.reshape(M, C), out_dtype=torch.float32,
                           z_out=y2).view(xs.shape)
        else:
            hpre = torch.addmm(b1b, h2, w1b.t())                                 # [M, 4C]
            hg = _gelu_bf16(hpre)
            y2 = torch.addmm(b2b, hg, w2b.t())                                   # [M, C] bf16, pre-gamma
            out = torch.empty(xs.shape, device=xs.device, dtype=torch.float32)
            _lib.check(lib.cnx_scale_residual(xs.data_ptr(), _code(xs), y2.data_ptr(), _lib.ptr(gf), out.data_ptr(), _code(out), M, C,
                                              _stream()), "cnx_scale_residual")
        if need_grad:
            ctx.save_for_backward(h2, hpre, hg, y2 if gf is not None else None, w1b, w2b, gf)
            ctx.w1, ctx.w2 = w1, w2
        ctx.M, ctx.C, ctx.xs_dtype, ctx.h_shape = M, C, xs.dtype, h.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        h2, hpre, hg, y2, w1b, w2b, gf = ctx.saved_tensors
        M, C = ctx.M, ctx.C
        nig = ctx.needs_input_grad
        want_p = any(nig[2:]) and not _INPUT_GRAD_ONLY
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        dxs = None
        if nig[0]:
            dxs = g if g.dtype == ctx.xs_dtype else g.to(ctx.xs_dtype)
        dhin = dw1 = db1 = dw2 = db2 = dgamma = None
        if nig[1] or want_p:
            dos = torch.empty(M, C, device=g.device, dtype=torch.bfloat16)
            ws = None
            if want_p:
                dgamma = torch.empty(C, device=g.device, dtype=torch.float32)
                db2 = torch.empty(C, device=g.device, dtype=torch.float32)
                db1 = torch.empty(4 * C, device=g.device, dtype=torch.float32)
                ws = torch.empty(lib.cnx_colsum_ws_floats(4 * C), device=g.device, dtype=torch.float32)
            _lib.check(lib.cnx_scale_residual_bwd(g.data_ptr(), _code(g), _lib.ptr(y2) if want_p else None, _lib.ptr(gf), dos.data_ptr(),
                                                  _lib.ptr(dgamma), _lib.ptr(db2), _lib.ptr(ws), M, C, _stream()),
                       "cnx_scale_residual_bwd")
            dhpre, dhin = _mlp_input_grads(lib, dos, hpre, ctx.w1, ctx.w2, w1b, w2b, db1, ws, M, C, nig[1])
            if dhin is not None:
                dhin = dhin.view(ctx.h_shape)
            if want_p:
                dw2 = _wgrad(dos, hg)
                dw1 = _wgrad(dhpre, h2)
                if gf is None:
                    dgamma = None
        return dxs, dhin, dw1, db1, dw2, db2, dgamma


class _LinearLib(torch.autograd.Function):
    """``x W^T + b`` on bf16 rows (the qkv / proj linears of the ViT attention, timm ``Attention.forward``): the library GEMM with
    the bias in its epilogue; backward = input-gradient GEMM, the weight gradient as a split-K batched GEMM (``_wgrad``: K =
    50 432 rows as one GEMM is a handful of output tiles) and the bias gradient as a deterministic column sum
    (``cnx_scale_residual_bwd`` with neither gamma nor y) instead of autograd's single long-K GEMM and an ATen reduction."""

    @staticmethod
    def forward(ctx, x, w, b):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        wb = _cached((w,), "bf16", lambda t: t.to(torch.bfloat16).contiguous())
        if _gemm_ok(x2, wb, "linear"):
            y = _gemm_nt(x2, wb, EPI_BIAS, bias=_f32(b))
        else:
            bb = _cached((b,), "bf16", lambda t: t.to(torch.bfloat16).contiguous())
            y = torch.addmm(bb, x2, wb.t())
        ctx.save_for_backward(x2, wb)
        ctx.x_shape, ctx.w = x.shape, w
        return y.view(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, wb = ctx.saved_tensors
        N = wb.shape[0]
        dy2 = dy.reshape(-1, N)
        if dy2.dtype != torch.bfloat16 or not dy2.is_contiguous():
            dy2 = dy2.to(torch.bfloat16).contiguous()
        M = dy2.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            if _gemm_dims_ok(dy2, N, wb.shape[1], "linear"):
                dx = _gemm_nt(dy2, _bf16_t(ctx.w), EPI_BIAS).view(ctx.x_shape)    # B = W^T [in, out]
            elif N % 64 != 0 and _gemm_dims_ok(dy2, (N + 63) // 64 * 64, wb.shape[1], "linear"):
                # a contraction length cnx_gemm_nt does not take (the 1000-class head: K = 1000): both operands zero-padded to the
                # next multiple of 64 - a few KB at the head's row counts - so that no GEMM of an attack pass is the library's
                Kp = (N + 63) // 64 * 64
                dyp = torch.zeros(M, Kp, device=dy2.device, dtype=torch.bfloat16)
                dyp[:, :N].copy_(dy2)
                wtp = _cached((ctx.w,), "bf16_t_pad64", lambda t: F.pad(t.to(torch.bfloat16).t(), (0, Kp - N)).contiguous())
                dx = _gemm_nt(dyp, wtp, EPI_BIAS).view(ctx.x_shape)
            else:
                dx = (dy2 @ wb).view(ctx.x_shape)
        dw = db = None
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _INPUT_GRAD_ONLY:
            dw = _wgrad(dy2, x2)
            db = torch.empty(N, device=dy2.device, dtype=torch.float32)
            zeros = torch.empty(N, device=dy2.device, dtype=torch.float32)
            ws = torch.empty(lib.cnx_colsum_ws_floats(N), device=dy2.device, dtype=torch.float32)
            _lib.check(lib.cnx_scale_residual_bwd(dy2.data_ptr(), _code(dy2), None, None, None, zeros.data_ptr(), db.data_ptr(),
                                                  ws.data_ptr(), M, N, _stream()), "cnx_scale_residual_bwd(colsum)")
        return dx, dw, db


_LINEAR_LIB = True


def linear_lib(x, w, b):
    """``F.linear(x, w, b)`` for a bf16 activation under bf16 autocast through ``_LinearLib``; the plain call otherwise."""
    if (MODE == "eager" or not _LINEAR_LIB or not x.is_cuda or x.dtype != torch.bfloat16 or b is None or not x.is_contiguous()
            or w.shape[0] % 4 != 0 or not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return F.linear(x, w, b)
    return _LinearLib.apply(x, w, b)


_MLP_RESIDUAL = True


def mlp_residual(xs, h, w1, b1, w2, b2, gamma=None):
    """``xs + gamma * fc2(GELU(fc1(h)))`` (fp32 result): ``_MlpResidual`` for a bf16 activation ``h`` under bf16 autocast, the
    plain composition otherwise."""
    C = h.shape[-1]
    if (MODE == "eager" or not _MLP_RESIDUAL or not h.is_cuda or h.dtype != torch.bfloat16 or not h.is_contiguous() or not xs.is_contiguous()
            or xs.dtype not in (torch.float32, torch.bfloat16) or xs.shape != h.shape or C % 8 != 0 or b1 is None or b2 is None
            or tuple(w1.shape) != (4 * C, C) or tuple(w2.shape) != (C, 4 * C)
            or not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return scale_residual(xs, F.linear(F.gelu(F.linear(h, w1, b1)), w2, b2), gamma)
    return _MlpResidual.apply(xs, h, w1, b1, w2, b2, gamma)


def _pack_mlp_bwd(w1, w2):
    """fc1 / fc2 weights -> the three operand-fragment sets of the fused backward (``cnx_mlp_pack_weights_bwd``)."""
    lib = _lib.load()
    C = w1.shape[1]
    w1, w2 = w1.contiguous(), w2.contiguous()
    if w1.dtype != w2.dtype or w1.dtype not in (torch.float32, torch.bfloat16):
        w1, w2 = w1.float(), w2.float()
    wb = torch.empty(lib.cnx_mlp_packed_bwd_elems(C), device=w1.device, dtype=torch.bfloat16)
    _lib.check(lib.cnx_mlp_pack_weights_bwd(w1.data_ptr(), w2.data_ptr(), _code(w1), wb.data_ptr(), C, _stream()),
               "cnx_mlp_pack_weights_bwd")
    return wb


class _BlockFused(torch.autograd.Function):
    """One ConvNeXt block on channels-last rows under bf16 autocast (``models/convnext.py:37-50``):

        x [N,H,W,C] -> u = dw7x7(x) (bf16)  ->  x + gamma * fc2(GELU(fc1(LN(u))))

    The depthwise stencil is its own kernel (``cnx_dwconv7x7_nhwc``); what follows it runs, by width and pass (round 6 state):

    * C = 128 ... 384, attack passes: ``cnx_block_mlp_fwd_hpre`` (LN prologue, two chained MFMA GEMMs, hidden activation on chip,
      bias / layer-scale / residual epilogue, Hpre tiles into a workspace) and ``cnx_block_mlp_bwd_input_hpre`` (dH, GELU', da and the
      LayerNorm backward in ONE kernel; at C = 256 / 384 both directions on wavefront pairs, ``blk2_fwd_kernel`` / ``blk2_bwd_kernel``);
    * C = 96, attack passes: ``cnx_block_mlp_fwd`` and the recomputing ``cnx_block_mlp_bwd_input`` (saved: x, u, the LN statistics);
    * C = 128 ... 384, training pass (row counts that are multiples of 64): ``cnx_block_mlp_fwd_train`` / ``cnx_block_mlp_bwd_train_hpre_ln``
      - H, Hpre, dHpre leave as accumulator-order tiles, both weight gradients (and d(b1), d(b2) as column sums) are ``cnx_gemm_tn_ex``
      contractions over those tiles, d(gamma) / d(ln_w) / d(ln_b) algebraic functions of them (``cnx_block_dgamma``, ``cnx_block_dln``,
      with direct sums for ill-conditioned channels); C = 96: the recomputing backward with the same emit (``cnx_block_mlp_bwd_acc_ln``);
    * C >= 512 (and every width under ``APGD_OPS`` / kernel-set fallbacks): LayerNorm, ``cnx_gemm_nt`` with fused epilogues (fc1 + bias +
      GELU + Hpre, fc2 + bias + layer scale + residual, dH with GELU') inside the attack, library GEMMs between the one-pass kernels in
      the training pass, weight gradients on ``cnx_gemm_tn``.

    Either way the residual gradient rides into the depthwise input-gradient
    kernel as its ``add`` operand, and parameter gradients are skipped entirely inside the attack."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_w, ln_b, eps, w1, b1, w2, b2, gamma, grad_mode=True):
        # grad_mode: torch.is_grad_enabled() of the CALLER (inside forward() it is always off)
        lib = _lib.load()
        N, H, W, C = x.shape
        M = N * H * W
        w49c = _cached((dw_w,), "w49c", lambda w: w.float().reshape(C, 49).t().contiguous())
        dwb = _f32(dw_b) if dw_b is not None else None
        lw, lb, b1f, b2f = _f32(ln_w), _f32(ln_b), _f32(b1), _f32(b2)
        gf = _f32(gamma) if gamma is not None else None
        u = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
        _lib.check(lib.cnx_dwconv7x7_nhwc(x.data_ptr(), _code(x), w49c.data_ptr(), _lib.ptr(dwb), None, u.data_ptr(),
                                          _code(u), N, H, W, C, 0, _stream()), "cnx_dwconv7x7_nhwc")
        out = torch.empty(x.shape, device=x.device,
                          dtype=torch.float32 if (gamma is not None or x.dtype == torch.float32) else x.dtype)
        need_grad = grad_mode and any(ctx.needs_input_grad)   # (needs_input_grad ignores torch.no_grad: parameters still 'require' it)
        need_p = need_grad and any(ctx.needs_input_grad[1:])
        fused = _use_fused_block(C) and bool(lib.cnx_block_mlp_bwd_supported(C))
        # The attack's passes (backward = input gradient only) run on the forward / input-gradient pair that hands Hpre over
        # through HBM instead of recomputing it (C = 128 ... 384: 15 - 19 % less time for the pair than fused forward +
        # recomputing backward; at C = 384 there is no recomputing backward and the plain fused forward also serves the passes
        # that need no backward at all)
        via_hpre = _use_hpre_block(C) and ((need_grad and _ATTACK_FWD) or ((not need_grad) and not fused))
        mean = rstd = y2 = a = hpre = h = None
        if need_p and not _ATTACK_FWD and not _INPUT_GRAD_ONLY and _use_train_hpre(C, M):
            # ---- training pass on the Hpre pair: the forward also leaves H (same tiles as Hpre) and the LN(u) rows for the weight gradients
            wf = _cached((w1, w2), "mlp_packed", _pack_mlp)
            mean = torch.empty(M, device=x.device, dtype=torch.float32)
            rstd = torch.empty(M, device=x.device, dtype=torch.float32)
            n_ws = lib.cnx_block_mlp_hpre_elems(M, C)
            hpre = torch.empty(n_ws, device=x.device, dtype=torch.bfloat16)
            h = torch.empty(n_ws, device=x.device, dtype=torch.bfloat16)
            a = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
            if gamma is not None and _dgamma_needs_y2(M, C, w2):
                y2 = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
            _lib.check(lib.cnx_block_mlp_fwd_train(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), eps, mean.data_ptr(), rstd.data_ptr(),
                                                   wf.data_ptr(), b1f.data_ptr(), b2f.data_ptr(), _lib.ptr(gf), x.data_ptr(), _code(x),
                                                   out.data_ptr(), _code(out), _lib.ptr(y2), hpre.data_ptr(), h.data_ptr(), a.data_ptr(),
                                                   M, C, _stream()), "cnx_block_mlp_fwd_train")
            wa = _cached((w1, w2), "mlp_packed_bwd", _pack_mlp_bwd)
            ctx.fused = "train_hpre"
            ctx.save_for_backward(x, w49c, u, mean, rstd, lw, lb, wa, None, b1f, gf, y2, a, hpre, h, b2f)
            ctx.has_dw_bias, ctx.eps, ctx.w1, ctx.w2 = dw_b is not None, eps, w1, w2
            return out
        if via_hpre:
            wf = _cached((w1, w2), "mlp_packed", _pack_mlp)
            if need_grad:
                mean = torch.empty(M, device=x.device, dtype=torch.float32)
                rstd = torch.empty(M, device=x.device, dtype=torch.float32)
                hpre = torch.empty(lib.cnx_block_mlp_hpre_elems(M, C), device=x.device, dtype=torch.bfloat16)
                _lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), eps, mean.data_ptr(),
                                                      rstd.data_ptr(), wf.data_ptr(), b1f.data_ptr(), b2f.data_ptr(), _lib.ptr(gf),
                                                      x.data_ptr(), _code(x), out.data_ptr(), _code(out), hpre.data_ptr(), M, C,
                                                      _stream()), "cnx_block_mlp_fwd_hpre")
                wa = _cached((w1, w2), "mlp_packed_bwd", _pack_mlp_bwd)
                ctx.fused = "hpre"
                ctx.save_for_backward(x, w49c, u, mean, rstd, lw, lb, wa, None, b1f, gf, None, None, hpre, None, b2f)
                ctx.has_dw_bias, ctx.eps = dw_b is not None, eps
            else:
                _lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), eps, None, None, wf.data_ptr(),
                                                 b1f.data_ptr(), b2f.data_ptr(), _lib.ptr(gf), x.data_ptr(), _code(x),
                                                 out.data_ptr(), _code(out), None, M, C, _stream()), "cnx_block_mlp_fwd")
            return out
        if need_grad or not fused:
            mean = torch.empty(M, device=x.device, dtype=torch.float32)
            rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        if fused:
            # LN + fc1 + GELU + fc2 + gamma + residual: ONE kernel, hidden activation on-chip (recomputed in the backward)
            wf = _cached((w1, w2), "mlp_packed", _pack_mlp)
            if need_p and gamma is not None and _dgamma_needs_y2(M, C, w2):
                y2 = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
            _lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), eps, _lib.ptr(mean), _lib.ptr(rstd),
                                             wf.data_ptr(), b1f.data_ptr(), b2f.data_ptr(), _lib.ptr(gf), x.data_ptr(), _code(x),
                                             out.data_ptr(), _code(out), _lib.ptr(y2), M, C, _stream()), "cnx_block_mlp_fwd")
            wa = _cached((w1, w2), "mlp_packed_bwd", _pack_mlp_bwd) if need_grad else None
            wb_ = None
        else:
            # widths without fused block kernels (384: the weight stream caps it below the library; 768): GEMMs in
            # hipBLASLt, everything around them in our one-pass kernels
            a = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
            _lib.check(lib.cnx_layernorm_fwd(u.data_ptr(), _code(u), lw.data_ptr(), lb.data_ptr(), eps, a.data_ptr(), _code(a),
                                             mean.data_ptr(), rstd.data_ptr(), M, C, 0, _stream()), "cnx_layernorm_fwd")
            wa = _cached((w1,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
            wb_ = _cached((w2,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
            b1b = _cached((b1,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
            b2b = _cached((b2,), "bf16", lambda w: w.to(torch.bfloat16).contiguous())
            if _gemm_ok(a, wa, "mlp") and _gemm_dims_ok(a, 4 * C, C, "mlp"):
                # fc1 + bias + GELU (+ Hpre for the backward) and fc2 + bias + gamma + residual: two kernels (cnx_gemm_nt)
                hpre = torch.empty(M, 4 * C, device=x.device, dtype=torch.bfloat16) if need_grad else None
                h = _gemm_nt(a, wa, EPI_BIAS_GELU, bias=b1f, z_out=hpre)
                y2 = torch.empty(M, C, device=x.device, dtype=torch.bfloat16) if need_p else None
                _lib.check(lib.cnx_gemm_nt(h.data_ptr(), 4 * C, wb_.data_ptr(), 4 * C, out.data_ptr(), C, _code(out), M, C, 4 * C,
                                           EPI_SCALE_RES, b2f.data_ptr(), _lib.ptr(gf), x.data_ptr(), C, _code(x), _lib.ptr(y2), None,
                                           C, _stream()), "cnx_gemm_nt")
            else:
                hpre = torch.addmm(b1b, a, wa.t())                               # [M, 4C]
                h = _gelu_bf16(hpre)
                y2 = torch.addmm(b2b, h, wb_.t())                                # [M, C] bf16, pre-gamma
                _lib.check(lib.cnx_scale_residual(x.data_ptr(), _code(x), y2.data_ptr(), _lib.ptr(gf), out.data_ptr(), _code(out),
                                                  M, C, _stream()), "cnx_scale_residual")
        if need_grad:
            ctx.fused = fused
            ctx.save_for_backward(x, w49c, u, mean, rstd, lw, lb, wa, wb_, b1f, gf, y2, a, hpre, h, b2f)
            ctx.has_dw_bias, ctx.eps, ctx.w1, ctx.w2 = dw_b is not None, eps, w1, w2
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w49c, u, mean, rstd, lw, lb, w1b, w2b, b1f, gf, y2, a_s, hpre, h, b2f = ctx.saved_tensors
        N, H, W, C = x.shape
        M = N * H * W
        nig = ctx.needs_input_grad
        want_p = any(nig[1:]) and not _INPUT_GRAD_ONLY
        g = g.contiguous()
        g2 = g.reshape(M, C)
        dw1 = db1 = dw2 = db2 = dgamma = None
        side = None                                                              # _SideLaunch of the weight-gradient work, joined at the end
        dln_id = False                                                           # d(ln_w), d(ln_b) from dW1 / d(b1) (cnx_block_dln)
        dlw_id = dlb_id = None                                                   # ... already computed (LayerNorm backward inside the block kernel)
        da = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)            # gradient w.r.t. LN(u)
        d_u = None
        if ctx.fused == "hpre":
            # ---- attack backward (C = 128 ... 384): Hpre from the forward's workspace, dH / GELU' / da / LayerNorm backward in ONE kernel
            if want_p:
                raise _lib.ApgdHipError("a block forward run under ops.attack_forward() can only be differentiated w.r.t. its input")
            if g2.dtype not in (torch.float32, torch.bfloat16):
                g2 = g2.float()
            d_u = da.view(u.shape)
            _lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g2.data_ptr(),
                                                        _code(g2), _lib.ptr(gf), w1b.data_ptr(), hpre.data_ptr(), d_u.data_ptr(),
                                                        M, C, _stream()), "cnx_block_mlp_bwd_input_hpre")
        elif ctx.fused == "train_hpre":
            # ---- training backward from the Hpre workspace: dH, GELU', da in one kernel that also leaves dO rows and dHpre tiles;
            #      both weight gradients (with their bias gradients as column sums) on cnx_gemm_tn_ex; d(gamma) in one pass over g, y2
            if g2.dtype not in (torch.float32, torch.bfloat16):
                g2 = g2.float()
            dos = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
            dhp = torch.empty(hpre.numel(), device=x.device, dtype=torch.bfloat16)
            w1p = getattr(ctx, "w1", None)
            ln_in = _LN_IN_TRAIN_BWD and want_p and w1p is not None and w1p.dtype == torch.float32 and w1p.is_contiguous() and M % 32 == 0
            if ln_in:
                # the LayerNorm backward in the kernel's epilogue (as in the attack's kernel): `da` comes out as d(loss)/du; the
                # LayerNorm's parameter gradients follow from dW1 / d(b1) below
                d_u = da.view(u.shape)
                _lib.check(lib.cnx_block_mlp_bwd_train_hpre_ln(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g2.data_ptr(),
                                                               _code(g2), _lib.ptr(gf), w1b.data_ptr(), hpre.data_ptr(), d_u.data_ptr(),
                                                               dos.data_ptr(), dhp.data_ptr(), M, C, _stream()),
                           "cnx_block_mlp_bwd_train_hpre_ln")
            else:
                _lib.check(lib.cnx_block_mlp_bwd_train_hpre(g2.data_ptr(), _code(g2), _lib.ptr(gf), w1b.data_ptr(), hpre.data_ptr(),
                                                            da.data_ptr(), dos.data_ptr(), dhp.data_ptr(), M, C, _stream()),
                           "cnx_block_mlp_bwd_train_hpre")
            if want_p:
                # (with the LayerNorm backward in the kernel above nothing below needs these results: they run on the side stream,
                #  under the depthwise gradients, and are joined in front of the return)
                side = _SideLaunch(x.device) if (_WGRAD_SIDE and ln_in) else None
                with side if side is not None else contextlib.nullcontext():
                    dw1, db1, dw2, db2 = _wgrad_block(dhp, a_s, h, dos, M, C)
                    dln_id = _DLN_FROM_DW1
                    if ln_in:
                        dlw_id, dlb_id = _block_dln(lib, w1p, dw1, db1, lw, lb, None, dhp, u, mean, rstd, M, C)
                    if gf is not None:
                        dgamma = _block_dgamma(lib, g2, y2, h, gf, ctx.w2, b2f, dw2, db2, M, C)
                if side is not None:
                    side.kept.extend((dos, dhp))
            del dos, dhp
        elif ctx.fused and not want_p:
            # ---- attack backward: ONE kernel down to the depthwise-conv output (LayerNorm backward in its epilogue)
            if g2.dtype not in (torch.float32, torch.bfloat16):
                g2 = g2.float()
            d_u = da.view(u.shape)
            _lib.check(lib.cnx_block_mlp_bwd_input(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   g2.data_ptr(), _code(g2), _lib.ptr(gf), w1b.data_ptr(), b1f.data_ptr(),
                                                   d_u.data_ptr(), M, C, _stream()), "cnx_block_mlp_bwd_input")
        elif ctx.fused:
            # ---- one kernel: LN recompute, dO = g*gamma, Hpre / dH / dHpre per hidden slice on-chip, da
            a = dos = ht = dhpt = None
            a_cols = C + 8 if _DB1_IN_GEMM else C
            acc_emit = want_p and _tn_ok(M, 4 * C, C) and _tn_ok(M, C, 4 * C)
            if acc_emit:
                # H and dHpre leave the kernel as accumulator-order tiles (two 16-byte stores per lane, no LDS transposition) and
                # the weight gradients - with d(b1), d(b2) as column sums - are cnx_gemm_tn_ex contractions over those tiles
                a = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
                dos = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
                ht = torch.empty(M * 4 * C, device=x.device, dtype=torch.bfloat16)
                dhpt = torch.empty(M * 4 * C, device=x.device, dtype=torch.bfloat16)
                if g2.dtype not in (torch.float32, torch.bfloat16):
                    g2 = g2.float()
                w1p = getattr(ctx, "w1", None)
                ln_in = _LN_IN_TRAIN_BWD and w1p is not None and w1p.dtype == torch.float32 and w1p.is_contiguous()
                if ln_in:
                    d_u = da.view(u.shape)
                fn = lib.cnx_block_mlp_bwd_acc_ln if ln_in else lib.cnx_block_mlp_bwd_acc
                _lib.check(fn(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                              g2.data_ptr(), _code(g2), _lib.ptr(gf), w1b.data_ptr(), b1f.data_ptr(),
                              da.data_ptr(), a.data_ptr(), dos.data_ptr(), ht.data_ptr(), dhpt.data_ptr(),
                              M, C, _stream()), "cnx_block_mlp_bwd_acc_ln" if ln_in else "cnx_block_mlp_bwd_acc")
                side = _SideLaunch(x.device) if (_WGRAD_SIDE and ln_in) else None
                with side if side is not None else contextlib.nullcontext():
                    dw1, db1, dw2, db2 = _wgrad_block(dhpt, a, ht, dos, M, C)
                    dln_id = _DLN_FROM_DW1
                    if ln_in:
                        dlw_id, dlb_id = _block_dln(lib, w1p, dw1, db1, lw, lb, None, dhpt, u, mean, rstd, M, C)
                    if gf is not None:
                        dgamma = _block_dgamma(lib, g2, y2, ht, gf, ctx.w2, b2f, dw2, db2, M, C)
                if side is not None:
                    side.kept.extend((a, dos, ht, dhpt))
                del a, dos, ht, dhpt
                want_emit = False
            else:
                want_emit = want_p
            if want_emit:
                # with 8 extra columns (1, 0, ..., 0) behind LN(u), the d(b1) sum is column C of the dW1 GEMM's result
                a = torch.empty(M, a_cols, device=x.device, dtype=torch.bfloat16)
                if a_cols != C:
                    a[:, C:] = _ones_col(x.device)
                dos = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
                ht = torch.empty(4 * C, M, device=x.device, dtype=torch.bfloat16)
                dhpt = torch.empty(4 * C, M, device=x.device, dtype=torch.bfloat16)
            if g2.dtype not in (torch.float32, torch.bfloat16):
                g2 = g2.float()
            if not acc_emit:
                _lib.check(lib.cnx_block_mlp_bwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                 g2.data_ptr(), _code(g2), _lib.ptr(gf), w1b.data_ptr(), b1f.data_ptr(),
                                                 da.data_ptr(), _lib.ptr(a), (a_cols if want_p else 0), _lib.ptr(dos), _lib.ptr(ht),
                                                 _lib.ptr(dhpt), M, C, _stream()), "cnx_block_mlp_bwd")
            if want_emit:
                dw1 = _wgrad_t(dhpt, a)                                          # [4C, C (+8)]
                if a_cols != C:
                    dw1, db1 = dw1[:, :C], dw1[:, C].contiguous()
                else:
                    db1 = dhpt.sum(1, dtype=torch.float32)
                dw2 = _wgrad_t(ht, dos).t()                                      # [C, 4C]
                # d(gamma) = sum_m g*y2 and d(b2) = sum_m dO in ONE pass over g and y2 (sums-only mode of the tail kernel;
                # as separate torch reductions they were a cast, a product and two sums: ~390 us per block at 56x56)
                dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
                db2 = torch.empty(C, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.cnx_colsum_ws_floats(C), device=x.device, dtype=torch.float32)
                y2p = y2.reshape(M, C).data_ptr() if y2 is not None else None
                _lib.check(lib.cnx_scale_residual_bwd(g2.data_ptr(), _code(g2), y2p, _lib.ptr(gf), None, dgamma.data_ptr(),
                                                      db2.data_ptr(), ws.data_ptr(), M, C, _stream()), "cnx_scale_residual_bwd")
                if gf is None:
                    dgamma = None
                del a, dos, ht, dhpt
        else:
            # ---- library GEMMs between the one-pass tails: dO (+ d(gamma), d(b2)), GELU' (+ d(b1)), split-K weight gradients
            if g2.dtype not in (torch.float32, torch.bfloat16):
                g2 = g2.float()
            dos = torch.empty(M, C, device=x.device, dtype=torch.bfloat16)
            ws = None
            if want_p:
                dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
                db2 = torch.empty(C, device=x.device, dtype=torch.float32)
                db1 = torch.empty(4 * C, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.cnx_colsum_ws_floats(4 * C), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_scale_residual_bwd(g2.data_ptr(), _code(g2), _lib.ptr(y2) if want_p else None, _lib.ptr(gf),
                                                  dos.data_ptr(), _lib.ptr(dgamma), _lib.ptr(db2), _lib.ptr(ws), M, C, _stream()),
                       "cnx_scale_residual_bwd")
            dhpre, da = _mlp_input_grads(lib, dos, hpre, ctx.w1, ctx.w2, w1b, w2b, db1, ws, M, C, True)   # da [M, C] bf16
            if want_p:
                dw2 = _wgrad(dos, h)
                dw1 = _wgrad(dhpre, a_s)
                if gf is None:
                    dgamma = None
            del dhpre, dos
        # ---- LayerNorm backward (already done by the fused input-gradient kernel)
        dlw, dlb, ws = dlw_id, dlb_id, None
        if d_u is None:
            d_u = torch.empty_like(u)
            w1p = getattr(ctx, "w1", None)
            dln_id = dln_id and want_p and w1p is not None and w1p.dtype == torch.float32 and w1p.is_contiguous()
            if want_p:
                dlw = torch.empty(C, device=x.device, dtype=torch.float32)
                dlb = torch.empty(C, device=x.device, dtype=torch.float32)
                if not dln_id:
                    ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_layernorm_bwd(da.data_ptr(), _code(da), u.data_ptr(), _code(u), lw.data_ptr(), None,
                                             mean.data_ptr(), rstd.data_ptr(), d_u.data_ptr(), _code(d_u),
                                             None if dln_id else _lib.ptr(dlw), None if dln_id else _lib.ptr(dlb), _lib.ptr(ws), M, C, 0, _stream()),
                       "cnx_layernorm_bwd")
            if dln_id:
                dlw, dlb = _block_dln(lib, w1p, dw1, db1, lw, lb, da, None, u, mean, rstd, M, C)
        # ---- depthwise conv backward; the residual branch's gradient rides along as the stencil's `add` input
        dx = None
        if nig[0]:
            # "+ add" (the residual gradient) is summed in fp32 inside the kernel; a bf16 block input (first block of a stage)
            # gets the sum rounded once
            dx = torch.empty_like(x)
            gadd = g if g.dtype == torch.float32 else g.float()
            _lib.check(lib.cnx_dwconv7x7_nhwc(d_u.data_ptr(), _code(d_u), w49c.data_ptr(), None, gadd.data_ptr(),
                                              dx.data_ptr(), _code(dx), N, H, W, C, 1, _stream()),
                       "cnx_dwconv7x7_nhwc(flip)")
        dww = dwb = None
        if want_p:
            g49 = torch.empty(49, C, device=x.device, dtype=torch.float32)
            dwb = torch.empty(C, device=x.device, dtype=torch.float32)
            ws2 = torch.empty(lib.cnx_dwconv7x7_wgrad_ws_floats(C), device=x.device, dtype=torch.float32)
            _lib.check(lib.cnx_dwconv7x7_wgrad_nhwc(x.data_ptr(), _code(x), d_u.data_ptr(), _code(d_u), g49.data_ptr(),
                                                    dwb.data_ptr(), ws2.data_ptr(), N, H, W, C, _stream()),
                       "cnx_dwconv7x7_wgrad_nhwc")
            dww = g49.t().reshape(C, 1, 7, 7)
            if not ctx.has_dw_bias:
                dwb = None
        if side is not None:
            side.join()
        return dx, dww, dwb, dlw, dlb, None, dw1, db1, dw2, db2, dgamma, None


def block_fused_supported(C):
    return bool(_lib.load().cnx_block_mlp_supported(C))


# Widths routed through the fused LN+MLP kernels (forward, input-gradient backward, training backward).  Measured on MI355X
# (tools/block_bench.py, batch 256): a clear win where the unfused block is HBM-bound (C = 96, 192; 128 and 256 - the first two
# stages of ConvNeXt-B - are instantiated from the same templates); at C = 384 the weight stream (128 rows per workgroup) caps
# it below the hipBLASLt composition, so that width stays on the library path for now.  APGD_BLOCK_FUSED overrides.
_FUSED_WIDTHS = os.environ.get("APGD_BLOCK_FUSED", "96,128,192,256")
_FUSED_WIDTHS = {int(v) for v in _FUSED_WIDTHS.split(",") if v.strip()}


# Widths whose attack passes (forward without backward; forward + input-gradient backward) use the fused kernels with the
# Hpre workspace while the training pass stays on the library GEMMs (cnx_block_mlp_fwd_hpre).  APGD_BLOCK_HPRE overrides ("" = none).
_HPRE_WIDTHS = os.environ.get("APGD_BLOCK_HPRE", "128,192,256,384")
_HPRE_WIDTHS = {int(v) for v in _HPRE_WIDTHS.split(",") if v.strip().isdigit()}


# Widths whose TRAINING pass runs on the Hpre kernel pair as well (round 5: cnx_block_mlp_fwd_train / cnx_block_mlp_bwd_train_hpre, weight
# gradients through cnx_gemm_tn_ex on the kernels' own tile layout) whenever the row count suits the contraction kernel (M % 64 == 0);
# otherwise - and with APGD_TRAIN_HPRE="" - the training pass of rounds 1 - 4 (recomputing fused kernels / library GEMMs).
_TRAIN_HPRE_WIDTHS = os.environ.get("APGD_TRAIN_HPRE", "128,192,256,384")
_TRAIN_HPRE_WIDTHS = {int(v) for v in _TRAIN_HPRE_WIDTHS.split(",") if v.strip().isdigit()}


def _use_train_hpre(C, M):
    return (C in _TRAIN_HPRE_WIDTHS and MODE != "eager" and bool(_lib.load().cnx_block_mlp_hpre_supported(C))
            and _tn_ok(M, 4 * C, C) and _tn_ok(M, C, 4 * C))


def _use_hpre_block(C):
    if MODE == "eager" or C not in _HPRE_WIDTHS:
        return False
    lib = _lib.load()
    return bool(lib.cnx_block_mlp_hpre_supported(C)) and bool(lib.cnx_block_mlp_supported(C))


def _use_fused_block(C):
    if MODE == "eager" or C not in _FUSED_WIDTHS:
        return False
    return block_fused_supported(C)


def convnext_block(x, dw_w, dw_b, ln_w, ln_b, eps, w1, b1, w2, b2, gamma):
    """``x + gamma * fc2(GELU(fc1(LN(dw7x7(x)))))`` on ``[N,C,H,W]`` (``models/convnext.py:37-50``)."""
    if MODE == "eager" or (x.is_cuda and not _hip_dtype_ok(x)):
        y = F.conv2d(x, dw_w, dw_b, padding=3, groups=x.shape[1]).permute(0, 2, 3, 1)
        y = F.layer_norm(y, ln_w.shape, ln_w, ln_b, eps)
    else:
        if not x.is_cuda:
            raise _lib.ApgdHipError("the fused ConvNeXt block needs a device tensor (no CPU path in the product)")
        xr = _rows(x)
        if xr.dtype not in (torch.float32, torch.bfloat16):
            xr = xr.float()
        if _act_dtype(xr) == torch.bfloat16 and x.shape[1] % 4 == 0:
            return _BlockFused.apply(xr, dw_w, dw_b, ln_w, ln_b, float(eps), w1, b1, w2, b2, gamma,
                                     torch.is_grad_enabled()).permute(0, 3, 1, 2)
        y = dwconv_ln(xr, dw_w, dw_b, ln_w, ln_b, eps)
    y = F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    if gamma is not None:
        y = y * gamma
    return x + y.permute(0, 3, 1, 2)


class _AttentionFused(torch.autograd.Function):
    """Packed ``[B,N,3C]`` bf16 projection -> ``[B,N,C]`` attention output through ``cnx_attention_fwd`` (one workgroup
    per (batch, head), K/V in LDS, scores in MFMA accumulators).  Saved for backward: qkv, the output and the per-row
    log-sum-exp; ``cnx_attention_bwd`` rebuilds the probabilities from them block by block."""

    @staticmethod
    def forward(ctx, qkv, num_heads, scale):
        lib = _lib.load()
        B, N, C3 = qkv.shape
        C = C3 // 3
        qkv = qkv.contiguous()
        out = torch.empty(B, N, C, device=qkv.device, dtype=torch.bfloat16)
        need_grad = ctx.needs_input_grad[0]
        lse = torch.empty(B, num_heads, N, device=qkv.device, dtype=torch.float32) if need_grad else None
        _lib.check(lib.cnx_attention_fwd(qkv.data_ptr(), out.data_ptr(), _lib.ptr(lse), B, N, num_heads, C // num_heads,
                                         float(scale), _stream()), "cnx_attention_fwd")
        if need_grad:
            ctx.save_for_backward(qkv, lse, out)
            ctx.num_heads, ctx.scale = num_heads, scale
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, lse, out = ctx.saved_tensors
        B, N, C3 = qkv.shape
        C, h = C3 // 3, ctx.num_heads
        lib = _lib.load()
        if lib.cnx_attention_bwd_supported(N, C // h):
            # two streaming kernels (dQ, then dK/dV): P is rebuilt block by block from the saved log-sum-exp
            dob = do.to(torch.bfloat16).contiguous()
            dqkv = torch.empty_like(qkv)
            dvec = torch.empty(B, h, N, device=qkv.device, dtype=torch.float32)
            _lib.check(lib.cnx_attention_bwd(qkv.data_ptr(), out.data_ptr(), dob.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                             dvec.data_ptr(), B, N, h, C // h, float(ctx.scale), _stream()), "cnx_attention_bwd")
            return dqkv, None, None
        # longer sequences (eval-time resolutions): library GEMMs on the rebuilt probabilities
        q, k, v = qkv.reshape(B, N, 3, h, C // h).permute(2, 0, 3, 1, 4).unbind(0)        # [B,h,N,d]
        do_ = do.reshape(B, N, h, C // h).permute(0, 2, 1, 3).to(torch.bfloat16)
        # scores in fp32: a bf16 GEMM output would round them (|s| ~ 10) by ~0.05, i.e. 5 % in the probabilities
        p = torch.exp((q.float() @ k.float().transpose(-2, -1)) * ctx.scale - lse.unsqueeze(-1))   # [B,h,N,N] fp32
        pb = p.to(torch.bfloat16)
        dv = pb.transpose(-2, -1) @ do_
        dp = (do_ @ v.transpose(-2, -1)).float()
        ds = (p * (dp - (dp * p).sum(-1, keepdim=True)) * ctx.scale).to(torch.bfloat16)
        dq = ds @ k
        dk = ds.transpose(-2, -1) @ q
        dqkv = torch.stack((dq, dk, dv), 0).permute(1, 3, 0, 2, 4).reshape(B, N, C3)
        return dqkv, None, None


def attention(qkv, num_heads, scale):
    """Multi-head softmax attention from a packed ``[B,N,3C]`` projection -> ``[B,N,C]``
    (timm 0.8 ``Attention.forward``)."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    if (MODE != "eager" and qkv.is_cuda and qkv.dtype == torch.bfloat16
            and _lib.load().cnx_attention_supported(N, C // num_heads)):
        return _AttentionFused.apply(qkv, num_heads, float(scale))
    q, k, v = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).unbind(0)
    a = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, C)


# ------------------------------------------------------------------------------ kernel sets (run-time A/B)
# The kernel-selection switches above are module globals read at call time (their APGD_* variables only set the start-up values), and
# the two that live in the library (APGD_BLK2, APGD_DW_SH) are behind cnx_runtime_switch: a RUNNING process can put the whole tree on
# another kernel set, drop its captured graphs and time both on one box - bench.py's interleaved A/B leg (`extra.ab`).
KERNEL_SETS = {
    # the tree as shipped
    "default": dict(wgrad="hip", stem_wgrad=True, train_hpre={128, 192, 256, 384}, dgamma=True, dln="dw1", fused_tracking=True,
                    blk2=3, pool_rows=True, dw_shared_halo=1, fwd_w8=0, blk2b=3, stem_ln_fused=False, attack_streams=2, gemm_auto_max=0,
                    tn_pair=True, share_derived=True, gemm_nt_tile=0,
                    tn_ring=1, wgrad_side=True),
    # the kernel set of the END OF ROUND 4 inside today's library (= APGD_WGRAD=lib APGD_STEM_WGRAD=lib APGD_TRAIN_HPRE="" APGD_DGAMMA=pass
    # APGD_DLN=pass APGD_FUSED_TRACKING=0 APGD_BLK2="" APGD_POOL_ROWS=0 APGD_DW_SH=0): library weight gradients, recomputing training
    # backward, the per-channel gradient passes, separate tracking pass, single-wavefront forward, round-4 depthwise strips
    # (library convolutions are never asked for a bias gradient any more - ops.conv_bias_grad: under hipGraph replay MIOpen's came back
    #  non-finite, which is how this set's first run found the hazard, gpurun_out/r6b - so the set runs the library stem gradients again)
    "round4": dict(wgrad="lib", stem_wgrad=False, train_hpre=set(), dgamma=False, dln="pass", fused_tracking=False, blk2=0,
                   pool_rows=False, dw_shared_halo=0, blk2b=0, tn_pair=False, share_derived=False, wgrad_side=False),
    # the end-of-round-5 selection: today's tree without round 6's wavefront-pair Hpre backward and paired weight-gradient launch
    "round5": dict(blk2b=0, tn_pair=False, share_derived=False, wgrad_side=False),
    # the weight-gradient work of a block's backward on the stream of the backward chain instead of a side stream
    "wgrad_main": dict(wgrad_side=False),
    # every graph rebuilds its own derived weight copies (packed / bf16 weights twice per step)
    "own_copies": dict(share_derived=False),
    # cnx_gemm_nt workgroup tile forced (0 = by the grid-size rule)
    "nt128": dict(gemm_nt_tile=1), "nt256x192": dict(gemm_nt_tile=2), "nt256": dict(gemm_nt_tile=3),
    # one launch per weight gradient (two cnx_gemm_tn_ex calls per block) instead of the paired launch
    "tn2": dict(tn_pair=False),
    # the paired launch on the stage loop of the single contractions / on a ring of 32-row stages instead of the early hand-over
    "tn2buf": dict(tn_ring=0), "tnring": dict(tn_ring=2),
    # single-switch experiments of round 6 (profiles/r06_ab.md)
    "stemln": dict(stem_ln_fused=True), "streams3": dict(attack_streams=3), "streams1": dict(attack_streams=1),
    # the C = 768 blocks of the TRAINING pass on cnx_gemm_nt with its fused epilogues instead of library GEMMs + one-pass tails
    "gemmtrain": dict(gemm_auto_max=40_000_000),
    # the wavefront-pair kernels also at C = 192 (both directions; a library built with -DBLK2_C192_BUILD=1, else nothing changes): slower
    "pair192": dict(blk2=7, blk2b=7),
    # round 6's measured negative: eight wavefronts (256 rows) per workgroup on one weight stream at C = 128 / 192
    # (profiles/r06_fused_mlp.md; needs a library built with -DBLK_FWD_W8_BUILD=1, else the switch reads back -1 and nothing changes)
    "w8": dict(fwd_w8=3),
}


def kernel_set(name_or_dict):
    """Select a kernel set in the running process; returns the settings that were in force (a dict ``kernel_set`` accepts).  The caller
    drops what was captured or cached under the old set (``graphed.reset()``, a fresh ``ATTrainStep``); results stay inside the parity
    bars either way - the sets differ in kernels and summation order, not in arithmetic."""
    global _WGRAD_MODE, STEM_WGRAD_HIP, _TRAIN_HPRE_WIDTHS, _DGAMMA_FROM_DW2, _DLN_FROM_DW1, _LN_IN_TRAIN_BWD, _POOL_ROWS, _STEM_LN_FUSED, _GEMM_AUTO_MAX
    global _TN_PAIR, SHARE_DERIVED, _WGRAD_SIDE
    from . import apgd as _apgd
    from . import graphed as _graphed
    new = KERNEL_SETS[name_or_dict] if isinstance(name_or_dict, str) else dict(name_or_dict)
    unknown = set(new) - set(KERNEL_SETS["default"])
    if unknown:
        raise ValueError(f"kernel_set: unknown switch(es) {sorted(unknown)}")
    lib = _lib.load()
    prev = dict(wgrad=_WGRAD_MODE, stem_wgrad=STEM_WGRAD_HIP, train_hpre=set(_TRAIN_HPRE_WIDTHS), dgamma=_DGAMMA_FROM_DW2,
                dln="dw1" if _LN_IN_TRAIN_BWD else ("kernel" if _DLN_FROM_DW1 else "pass"), fused_tracking=_apgd.FUSED_TRACKING,
                blk2=int(lib.cnx_runtime_switch(0, -1)), pool_rows=_POOL_ROWS, dw_shared_halo=int(lib.cnx_runtime_switch(1, -1)),
                fwd_w8=max(0, int(lib.cnx_runtime_switch(2, -1))), blk2b=int(lib.cnx_runtime_switch(3, -1)),
                stem_ln_fused=_STEM_LN_FUSED, attack_streams=_graphed.STREAMS, gemm_auto_max=_GEMM_AUTO_MAX, tn_pair=_TN_PAIR,
                share_derived=SHARE_DERIVED, gemm_nt_tile=int(lib.cnx_runtime_switch(4, -1)),
                tn_ring=int(lib.cnx_runtime_switch(5, -1)), wgrad_side=_WGRAD_SIDE)
    if "wgrad_side" in new:
        _WGRAD_SIDE = bool(new["wgrad_side"])
    if "gemm_nt_tile" in new:
        lib.cnx_runtime_switch(4, int(new["gemm_nt_tile"]) & 3)
    if "tn_ring" in new:
        lib.cnx_runtime_switch(5, int(new["tn_ring"]) & 3)
    if "share_derived" in new:
        SHARE_DERIVED = bool(new["share_derived"])
    if "tn_pair" in new:
        _TN_PAIR = bool(new["tn_pair"])
    if "wgrad" in new:
        if new["wgrad"] not in ("hip", "lib"):
            raise ValueError(f"kernel_set: wgrad={new['wgrad']!r}")
        _WGRAD_MODE = new["wgrad"]
    if "stem_wgrad" in new:
        STEM_WGRAD_HIP = bool(new["stem_wgrad"])
    if "train_hpre" in new:
        _TRAIN_HPRE_WIDTHS = {int(v) for v in new["train_hpre"]}
    if "dgamma" in new:
        _DGAMMA_FROM_DW2 = bool(new["dgamma"])
    if "dln" in new:
        if new["dln"] not in ("dw1", "kernel", "pass"):
            raise ValueError(f"kernel_set: dln={new['dln']!r}")
        _DLN_FROM_DW1, _LN_IN_TRAIN_BWD = new["dln"] != "pass", new["dln"] == "dw1"
    if "fused_tracking" in new:
        _apgd.FUSED_TRACKING = bool(new["fused_tracking"])
    if "pool_rows" in new:
        _POOL_ROWS = bool(new["pool_rows"])
    if "blk2" in new:
        lib.cnx_runtime_switch(0, int(new["blk2"]) & 7)
    if "dw_shared_halo" in new:
        lib.cnx_runtime_switch(1, 1 if new["dw_shared_halo"] else 0)
    if "fwd_w8" in new:
        lib.cnx_runtime_switch(2, int(new["fwd_w8"]) & 3)
    if "blk2b" in new:
        lib.cnx_runtime_switch(3, int(new["blk2b"]) & 7)
    if "stem_ln_fused" in new:
        _STEM_LN_FUSED = bool(new["stem_ln_fused"])
    if "attack_streams" in new:
        _graphed.STREAMS = max(1, int(new["attack_streams"]))
    if "gemm_auto_max" in new:
        _GEMM_AUTO_MAX = int(new["gemm_auto_max"])
    invalidate_weight_cache()
    return prev
