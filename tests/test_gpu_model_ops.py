"""GPU tests of the model-side HIP kernels (include/convnext_hip.h) against plain fp32 torch on the
CPU (oracle/models_ref.py restates the reference's modules; F.conv2d / F.layer_norm are the
fp32 reference of each floating-point kernel).  Tolerances are written next to each check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R
    assert torch.cuda.is_available()
    R._lib.load()
    return R


def S():
    return torch.cuda.current_stream().cuda_stream


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("N,C,H,W", [(2, 96, 14, 14), (1, 192, 7, 7), (2, 384, 9, 11), (3, 8, 5, 20), (1, 768, 7, 7),
                                     (2, 200, 6, 8), (1, 96, 56, 56), (6, 64, 14, 14), (5, 32, 7, 7), (3, 64, 28, 28),
                                     (2, 32, 30, 27), (1, 192, 80, 80), (1, 384, 40, 40), (2, 384, 4, 4), (2, 64, 20, 20), (1, 32, 24, 24),
                                     (2, 32, 10, 10), (2, 192, 28, 28), (3, 384, 14, 14), (2, 768, 7, 7), (3, 96, 28, 28), (2, 96, 35, 21),
                                     (9, 96, 56, 56), (2, 64, 10, 14), (1, 96, 9, 7), (4, 128, 33, 28), (1, 64, 3, 7), (2, 32, 1, 14)])
# cfg#4 / cfg#1 maps; 20x20: output tile > input tile; 35x21x96: an odd strip count (idle half-wave); 10x14, 9x7, 33x28: widths the LDS-DMA
# form takes with heights that are no multiple of its 7-row groups (surplus steps store nothing) and several bands per image; 3x7, 1x14:
# maps lower than the window
@pytest.mark.parametrize("xdt,odt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("win", [0, 1], ids=["lds-ring", "dma-window"])
def test_dwconv7x7_fwd_bwd_data_wgrad(R, N, C, H, W, xdt, odt, win):
    """All-fp32 calls are exact fp32 stencils.  Any call with a bf16 operand is the autocast convolution: input AND
    filter rounded to bf16, fp32 accumulation (packed bf16 dot products) - the references below round the same way.
    ``win``: both kernel families on every shape (cnx_dwconv7x7_win_policy 0 / 1: LDS ring; the sliding-window kernels, ragged
    widths included)."""
    lib = R._lib.load()
    prev = lib.cnx_dwconv7x7_win_policy(win)
    try:
        _dwconv_case(R, lib, N, C, H, W, xdt, odt)
    finally:
        lib.cnx_dwconv7x7_win_policy(prev)


def _dwconv_case(R, lib, N, C, H, W, xdt, odt):
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, 7, 7, generator=g) * 0.2
    b = torch.randn(C, generator=g)
    add = torch.randn(N, H, W, C, generator=g)
    bf = lambda t: t.to(torch.bfloat16).float()

    def operands(in_dt, out_dt, t):                          # what the kernel multiplies for this dtype pair
        mixed = in_dt != torch.float32 or out_dt != torch.float32
        return (bf(t) if mixed else t.to(in_dt).float()), (bf(w) if mixed else w)

    xq = x.to(xdt).float()                                   # the stored input
    x_rows = xq.permute(0, 2, 3, 1).contiguous().to(xdt).cuda()
    w49c = w.reshape(C, 49).t().contiguous().cuda()
    out = torch.empty(N, H, W, C, device="cuda", dtype=odt)
    code = R._lib.dtype_code
    assert lib.cnx_dwconv7x7_nhwc(x_rows.data_ptr(), code(xdt), w49c.data_ptr(), b.cuda().data_ptr(), None,
                                  out.data_ptr(), code(odt), N, H, W, C, 0, S()) == 0
    xm, wm = operands(xdt, odt, xq)
    ref = F.conv2d(xm, wm, b, padding=3, groups=C)
    tol = 1e-4 if odt == torch.float32 else 2e-2              # bf16 output: 2^-8 relative
    close(out.permute(0, 3, 1, 2), ref, tol, tol)
    # fused "+ add", fp32 output
    addc = add.cuda()
    out2 = torch.empty(N, H, W, C, device="cuda", dtype=torch.float32)
    assert lib.cnx_dwconv7x7_nhwc(x_rows.data_ptr(), code(xdt), w49c.data_ptr(), None, addc.data_ptr(), out2.data_ptr(),
                                  0, N, H, W, C, 0, S()) == 0
    xm, wm = operands(xdt, torch.float32, xq)
    close(out2.permute(0, 3, 1, 2), F.conv2d(xm, wm, None, padding=3, groups=C) + add.permute(0, 3, 1, 2), 1e-4, 1e-4)
    # input gradient = same kernel with the rotated filter
    dy = torch.randn(N, C, H, W, generator=g)
    dyq = dy.to(odt).float()
    dy_rows = dyq.permute(0, 2, 3, 1).contiguous().to(odt).cuda()
    dx = torch.empty(N, H, W, C, device="cuda", dtype=torch.float32)
    assert lib.cnx_dwconv7x7_nhwc(dy_rows.data_ptr(), code(odt), w49c.data_ptr(), None, None, dx.data_ptr(), 0, N, H, W,
                                  C, 1, S()) == 0
    dym, wm = operands(odt, torch.float32, dyq)
    xr = xq.clone().requires_grad_()
    (gx,) = torch.autograd.grad(F.conv2d(xr, wm, None, padding=3, groups=C), xr, dym)
    close(dx.permute(0, 3, 1, 2), gx, 1e-4, 1e-4)
    # filter / bias gradient kernel: with a bf16 dy (autocast) an fp32 x is rounded to bf16 like the convolution did
    # (packed-dot kernel, C % 32 == 0); otherwise fp32 arithmetic on the stored operands
    wr = w.clone().requires_grad_()
    br = b.clone().requires_grad_()
    xw = bf(xq) if (odt == torch.bfloat16 and C % 32 == 0) else xq
    gw, gb = torch.autograd.grad(F.conv2d(xw, wr, br, padding=3, groups=C), (wr, br), dyq)
    dw = torch.empty(49, C, device="cuda")
    db = torch.empty(C, device="cuda")
    ws = torch.empty(lib.cnx_dwconv7x7_wgrad_ws_floats(C), device="cuda")
    assert lib.cnx_dwconv7x7_wgrad_nhwc(x_rows.data_ptr(), code(xdt), dy_rows.data_ptr(), code(odt), dw.data_ptr(),
                                        db.data_ptr(), ws.data_ptr(), N, H, W, C, S()) == 0
    scale = float(gw.abs().max()) + 1e-6
    close(dw.t().reshape(C, 1, 7, 7), gw, 1e-4, 2e-5 * scale + 1e-5)       # fp32 accumulation, different order
    close(db, gb, 1e-4, 1e-4 * float(gb.abs().max()) + 1e-5)


@pytest.mark.parametrize("N,C,H,W", [(128, 128, 80, 80), (128, 768, 10, 10), (64, 64, 24, 64), (96, 192, 32, 32), (128, 96, 24, 64), (96, 96, 32, 35),
                                     (64, 160, 40, 22)])
# (the last three: 32-channel wavefronts with the shared column halo - a ragged last strip, an odd strip count = an idle half-wavefront)
def test_dwconv_dma_ragged_fp32_stores_under_load(R, N, C, H, W):
    """bf16 in, fp32 out, no add operand, widths with W % 7 in 1..4, the whole chip busy: the second 16-byte store group of the
    ragged last strip has no active lane in some wavefronts; the kernel must still issue a constant number of VMEM instructions
    per step (a sink load stands in), or its counted vmcnt waits let a row be read before its LDS-DMA has landed (round-4 advice).
    Checked against the fp32 convolution of the same bf16-rounded operands on the GPU (library path, different kernel), twice."""
    lib = R._lib.load()
    prev = lib.cnx_dwconv7x7_win_policy(1)
    try:
        g = torch.Generator(device="cuda").manual_seed(C + W)
        dy = torch.randn(N, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, 1, 7, 7, device="cuda", generator=g) * 0.2)
        w49c = w.reshape(C, 49).t().contiguous()
        code = R._lib.dtype_code
        ref = F.conv2d(dy.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float().flip(-1, -2), None, padding=3, groups=C)
        for _ in range(2):
            dx = torch.full((N, H, W, C), float("nan"), device="cuda")
            assert lib.cnx_dwconv7x7_nhwc(dy.data_ptr(), code(torch.bfloat16), w49c.data_ptr(), None, None, dx.data_ptr(), 0,
                                          N, H, W, C, 1, S()) == 0
            err = (dx.permute(0, 3, 1, 2) - ref).abs().max().item()
            assert err <= 1e-3, err                               # fp32 accumulation of 49 bf16 products, |values| ~ 1
    finally:
        lib.cnx_dwconv7x7_win_policy(prev)


@pytest.mark.parametrize("M_,C", [(5, 48), (37, 96), (64, 64), (10, 144), (33, 192), (21, 384), (9, 768), (4, 1536), (3, 8),
                                  (1000, 48), (3000, 96), (700, 384), (300, 768)])
@pytest.mark.parametrize("gelu", [0, 1])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm_rows_fwd_bwd(R, M_, C, gelu, dt):
    lib = R._lib.load()
    code = R._lib.dtype_code
    g = torch.Generator().manual_seed(M_ * 31 + C)
    x = (torch.randn(M_, C, generator=g) * 2 + 0.3).to(dt).float()
    w = torch.randn(C, generator=g) * 0.5 + 1
    b = torch.randn(C, generator=g) * 0.5
    dy = torch.randn(M_, C, generator=g).to(dt).float()
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    ref = F.layer_norm(xr, (C,), wr, br, 1e-6)
    if gelu:
        ref = F.gelu(ref)
    gx, gw, gb = torch.autograd.grad(ref, (xr, wr, br), dy)
    xd, wd, bd, dyd = x.to(dt).cuda(), w.cuda(), b.cuda(), dy.to(dt).cuda()
    y = torch.empty(M_, C, device="cuda", dtype=dt)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    assert lib.cnx_layernorm_fwd(xd.data_ptr(), code(dt), wd.data_ptr(), bd.data_ptr(), 1e-6, y.data_ptr(), code(dt),
                                 mean.data_ptr(), rstd.data_ptr(), M_, C, gelu, S()) == 0
    tol = 2e-5 if dt == torch.float32 else 1.6e-2
    close(y, ref, tol, tol)
    close(mean, x.mean(1), 1e-5, 1e-5)
    close(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6), 1e-4, 1e-5)
    dx = torch.empty(M_, C, device="cuda", dtype=dt)
    dw, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ws = torch.empty(lib.cnx_layernorm_bwd_ws_floats(C), device="cuda")
    assert lib.cnx_layernorm_bwd(dyd.data_ptr(), code(dt), xd.data_ptr(), code(dt), wd.data_ptr(), bd.data_ptr(),
                                 mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), code(dt), dw.data_ptr(), db.data_ptr(),
                                 ws.data_ptr(), M_, C, gelu, S()) == 0
    close(dx, gx, 5 * tol, 5 * tol)
    close(dw, gw, 1e-4, 1e-4 * float(gw.abs().max()) + 1e-5)
    close(db, gb, 1e-4, 1e-4 * float(gb.abs().max()) + 1e-5)
    # input-gradient only (the attack's backward): parameter outputs may be NULL
    dx2 = torch.empty_like(dx)
    assert lib.cnx_layernorm_bwd(dyd.data_ptr(), code(dt), xd.data_ptr(), code(dt), wd.data_ptr(), bd.data_ptr(),
                                 mean.data_ptr(), rstd.data_ptr(), dx2.data_ptr(), code(dt), None, None, None, M_, C, gelu,
                                 S()) == 0
    assert torch.equal(dx, dx2)


def _grads(out, inputs, cot):
    return torch.autograd.grad(out, inputs, cot, allow_unused=True)


@pytest.mark.parametrize("C,H", [(96, 14), (192, 7), (384, 6), (128, 14), (256, 7)])
@pytest.mark.parametrize("gamma", [True, False])
def test_convnext_block_matches_reference_block(R, C, H, gamma):
    torch.manual_seed(C + H)
    ref = M.CNBlock(C, ls_init=0.5 if gamma else 0).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.2)
    x = torch.randn(2, C, H, H)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    cot = torch.randn_like(yr)
    params = list(ref.parameters())
    gref = _grads(yr, [xr] + params, cot)
    # product block, fp32 (no autocast)
    blk = R.architecture.ConvNeXtBlock(C, ls_init_value=0.5 if gamma else 0).cuda().eval()
    blk.load_state_dict(ref.state_dict())
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    yd = blk(xd)
    gd = _grads(yd, [xd] + list(blk.parameters()), cot.cuda())
    close(yd, yr, 2e-4, 2e-4)
    for a, b, (n, _) in zip(gd, gref, [("x", None)] + list(ref.named_parameters())):
        close(a, b, 2e-3, 2e-4 * float(b.abs().max()) + 1e-5)
    # bf16 autocast: activations bf16 / fp32 accumulation; bar = bf16 <= 1e-2 relative (north_star)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xd2 = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        yb = blk(xd2)
    assert yb.dtype == torch.float32 if gamma else True
    gall = _grads(yb, [xd2] + list(blk.parameters()), cot.cuda())
    gb = gall[0]
    for a, b, (n, _) in zip(gall[1:], gref[1:], ref.named_parameters()):        # parameter gradients (training backward)
        rel = float((a.float().cpu() - b).norm() / b.norm())
        assert rel < 2e-2, (n, rel)
    res_ref = (yr - x).detach()
    res_dev = (yb.float().cpu() - x).detach()
    assert float((res_dev - res_ref).norm() / res_ref.norm()) < 1e-2
    assert float((gb.float().cpu() - gref[0]).norm() / gref[0].norm()) < 1.5e-2


@pytest.mark.parametrize("shape,N", [((3, 197, 768), 2304), ((256, 197, 384), 384), ((2, 5, 64), 36)])
def test_linear_lib_vs_autograd_linear(R, shape, N):
    """``ops.linear_lib`` (qkv / proj of the ViT attention): same output as ``F.linear`` under autocast, input / weight / bias
    gradients vs fp32 autograd on the bf16-quantised operands (the weight gradient is a split-K batched GEMM, the bias
    gradient a deterministic column sum); attack mode skips the parameter gradients."""
    torch.manual_seed(N)
    K = shape[-1]
    x = torch.randn(*shape, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).requires_grad_()
    b = (0.2 * torch.randn(N, device="cuda")).requires_grad_()
    g = torch.randn(*shape[:-1], N, device="cuda").to(torch.bfloat16)
    xd = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = R.ops.linear_lib(xd, w, b)
        y0 = F.linear(x, w, b)
    assert isinstance(y.grad_fn, R.ops._LinearLib._backward_cls) and y.dtype == torch.bfloat16
    assert float((y.float() - y0.float()).abs().max()) <= 0.05 * float(y0.float().abs().max())
    gx, gw, gb = torch.autograd.grad(y, [xd, w, b], g, retain_graph=True)
    xr = x.float().requires_grad_()
    wr, br = w.detach().to(torch.bfloat16).float().requires_grad_(), b.detach().clone().requires_grad_()
    rx, rw, rb = torch.autograd.grad(F.linear(xr, wr, br), [xr, wr, br], g.float())
    rel_ = lambda a_, b_: float((a_.float() - b_).norm() / (b_.norm() + 1e-30))
    assert gw.dtype == torch.float32 and gb.dtype == torch.float32
    assert rel_(gx, rx) < 1e-2 and rel_(gw, rw) < 1e-2 and rel_(gb, rb) < 1e-3, (rel_(gx, rx), rel_(gw, rw), rel_(gb, rb))
    with R.ops.input_grad_only():
        (ga,) = torch.autograd.grad(y, xd, g)
    assert torch.equal(ga, gx)


@pytest.mark.parametrize("gamma", [True, False])
@pytest.mark.parametrize("shape", [(3, 197, 768), (2, 50, 384), (4, 16, 96)])
def test_mlp_residual_vs_fp32_reference(R, shape, gamma):
    """``ops.mlp_residual`` (second half of a ViT block: xs + ls2(fc2(GELU(fc1(h))))) on library GEMMs + cnx_gelu_fwd /
    cnx_scale_residual[_bwd] / cnx_gelu_bwd_colsum vs fp32 autograd of the same bf16-quantised operands: output, both input
    gradients, all parameter gradients; attack mode (input gradients only); cnx_gelu_fwd vs torch GELU."""
    torch.manual_seed(sum(shape) + int(gamma))
    C = shape[-1]
    xs = torch.randn(*shape, device="cuda")
    h = torch.randn(*shape, device="cuda").to(torch.bfloat16)
    w1 = (torch.randn(4 * C, C, device="cuda") * C ** -0.5).requires_grad_()
    w2 = (torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5).requires_grad_()
    b1, b2 = (0.2 * torch.randn(4 * C, device="cuda")).requires_grad_(), (0.2 * torch.randn(C, device="cuda")).requires_grad_()
    gm = (0.5 + 0.2 * torch.randn(C, device="cuda")).requires_grad_() if gamma else None
    g = torch.randn(*shape, device="cuda")
    q = lambda t: t.detach().to(torch.bfloat16).float()
    # fp32 reference on the bf16-quantised operands
    xr, hr = xs.clone().requires_grad_(), h.float().requires_grad_()
    pr = [q(w1).requires_grad_(), q(b1).requires_grad_(), q(w2).requires_grad_(), q(b2).requires_grad_()]
    y = F.linear(F.gelu(F.linear(hr, pr[0], pr[1])), pr[2], pr[3])
    gr = gm.detach().clone().requires_grad_() if gamma else None
    ref = xr + (y * gr if gamma else y)
    gref = torch.autograd.grad(ref, [xr, hr] + pr + ([gr] if gamma else []), g)
    xd, hd = xs.clone().requires_grad_(), h.clone().requires_grad_()
    params = [w1, b1, w2, b2] + ([gm] if gamma else [])
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = R.ops.mlp_residual(xd, hd, w1, b1, w2, b2, gm)
    assert out.dtype == torch.float32 and isinstance(out.grad_fn, R.ops._MlpResidual._backward_cls)
    got = torch.autograd.grad(out, [xd, hd] + params, g, retain_graph=True)
    rel_ = lambda a_, b_: float((a_.float() - b_).norm() / (b_.norm() + 1e-30))
    assert rel_(out - xs, (ref - xr).detach()) < 1e-2
    assert torch.equal(got[0], g)
    bars = [2e-2] * 6
    for a_, b_, bar in zip(got[1:], gref[1:], bars):
        assert rel_(a_, b_) < bar, (rel_(a_, b_), a_.shape)
    with R.ops.input_grad_only():
        ga = torch.autograd.grad(out, [xd, hd], g)
    assert torch.equal(ga[1], got[1])
    # the stand-alone GELU kernel
    t = torch.randn(4096 * 8, device="cuda").mul_(3).to(torch.bfloat16)
    yk = R.ops._gelu_bf16(t)
    assert float((yk.float() - F.gelu(t.float())).abs().max()) < 2e-2 and float((yk.float() - F.gelu(t).float()).abs().mean()) < 1e-4
    lib = R._lib.load()
    assert lib.cnx_gelu_fwd(t.data_ptr(), yk.data_ptr(), 0, S()) == 0 and lib.cnx_gelu_fwd(t.data_ptr(), yk.data_ptr(), 12, S()) < 0
    assert lib.cnx_gelu_fwd(None, yk.data_ptr(), 16, S()) < 0


@pytest.mark.parametrize("shape", [(3, 197, 768), (2, 50, 384), (5, 7, 96), (1, 3, 40)])
def test_layer_norm_skip_sums_both_gradients_of_x_in_the_backward_kernel(R, shape):
    """``ops.layer_norm_skip`` (ViT blocks: x feeds norm(x) and the skip connection): outputs equal the plain LayerNorm op and x,
    and d(loss)/dx = skip gradient + LayerNorm backward comes out of ONE kernel (``cnx_layernorm_bwd_add``), bit-identical to the
    two-step sum, parameter gradients unchanged; a missing branch / skip gradient and the no-grad path are handled."""
    torch.manual_seed(sum(shape))
    C = shape[-1]
    x = torch.randn(*shape, device="cuda")
    w, b = (1 + 0.2 * torch.randn(C, device="cuda")).requires_grad_(), (0.2 * torch.randn(C, device="cuda")).requires_grad_()
    gy = torch.randn(*shape, device="cuda").to(torch.bfloat16)
    gs = torch.randn(*shape, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xa = x.clone().requires_grad_()
        y0 = R.ops.layer_norm_last(xa, w, b, 1e-6)
        g0 = torch.autograd.grad([y0, xa * 1.0], [xa, w, b], [gy, gs])
        xb = x.clone().requires_grad_()
        y1, xs = R.ops.layer_norm_skip(xb, w, b, 1e-6)
        assert y1.dtype == torch.bfloat16 and torch.equal(y1, y0) and torch.equal(xs, xb)
        g1 = torch.autograd.grad([y1, xs * 1.0], [xb, w, b], [gy, gs])
        for a_, b_ in zip(g0, g1):
            assert torch.equal(a_, b_)
        # only one of the two outputs used
        y2, xs2 = R.ops.layer_norm_skip(xb, w, b, 1e-6)
        (g2,) = torch.autograd.grad(xs2 * 1.0, xb, gs)
        assert torch.equal(g2, gs)
        y3, xs3 = R.ops.layer_norm_skip(xb, w, b, 1e-6)
        (g3,) = torch.autograd.grad(y3, xb, gy)
        (g3r,) = torch.autograd.grad(R.ops.layer_norm_last(xa, w, b, 1e-6), xa, gy)
        assert torch.equal(g3, g3r)
        with torch.no_grad():
            y4, xs4 = R.ops.layer_norm_skip(x, w, b, 1e-6)
        assert xs4 is x and torch.equal(y4, y0)
    # fp32 reference
    xr = x.clone().requires_grad_()
    yr = F.layer_norm(xr, (C,), w.detach(), b.detach(), 1e-6)
    (gr,) = torch.autograd.grad([yr, xr * 1.0], xr, [gy.float(), gs])
    assert float((g1[0] - gr).norm() / gr.norm()) < 1e-5
    # the C ABI refuses a skip gradient for a bf16 result
    lib = R._lib.load()
    M_ = x.numel() // C
    if C % 4 == 0:
        t = torch.empty(M_, device="cuda")
        rc = lib.cnx_layernorm_bwd_add(gy.data_ptr(), 1, x.data_ptr(), 0, w.data_ptr(), b.data_ptr(), t.data_ptr(), t.data_ptr(), gs.data_ptr(),
                                       torch.empty_like(gy).data_ptr(), 1, None, None, None, M_, C, 0, S())
        assert rc < 0


@pytest.mark.parametrize("C", [384, 192, 128])
@pytest.mark.parametrize("gamma", [True, False])
@pytest.mark.parametrize("N,H", [(2, 14), (3, 9), (1, 5)])
def test_block_c384_attack_passes_use_the_hpre_kernels_and_match_the_reference_block(R, gamma, N, H, C):
    """C = 384 (9 of ConvNeXt-T's 18 blocks), 192, 128: the attack's passes run cnx_block_mlp_fwd[_hpre] /
    cnx_block_mlp_bwd_input_hpre (forward writes Hpre, the input-gradient kernel reads it back), the training pass stays on the
    library GEMMs (384) or the recomputing fused pair (192, 128).  All three against the fp32 reference block; a parameter
    gradient asked of an attack-forward graph is refused."""
    assert R._lib.load().cnx_block_mlp_hpre_supported(C) == 1 and R.ops._use_hpre_block(C)
    torch.manual_seed(7 * N + H)
    ref = M.CNBlock(C, ls_init=0.5 if gamma else 0).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.2)
    x = torch.randn(N, C, H, H)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    cot = torch.randn_like(yr)
    (gref,) = torch.autograd.grad(yr, xr, cot)
    blk = R.architecture.ConvNeXtBlock(C, ls_init_value=0.5 if gamma else 0).cuda().eval()
    blk.load_state_dict(ref.state_dict())

    def relerr(a, b):
        return float((a.detach().float().cpu() - b.detach()).norm() / b.detach().norm())

    res_ref = (yr - x).detach()
    # attack pass: forward under attack_forward, backward under input_grad_only
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16), R.ops.attack_forward():
        ya = blk(xd)
    with R.ops.input_grad_only():
        (ga,) = torch.autograd.grad(ya, xd, cot.cuda(), retain_graph=True)
    assert relerr(ya.float().cpu() - x, res_ref) < 1e-2 and relerr(ga, gref) < 1.5e-2
    with pytest.raises(R._lib.ApgdHipError):
        torch.autograd.grad(ya, [xd] + list(blk.parameters()), cot.cuda())
    # forward without a backward (the attack's last iteration)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        yn = blk(x.cuda().contiguous(memory_format=torch.channels_last))
    assert torch.equal(yn, ya.detach())
    # training pass (library GEMMs): same numbers up to bf16 noise, all gradients available
    xt = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yt = blk(xt)
    gt = torch.autograd.grad(yt, [xt] + list(blk.parameters()), cot.cuda())
    assert relerr(yt.float().cpu() - x, res_ref) < 1e-2 and relerr(gt[0], gref) < 1.5e-2
    assert relerr(ga, gt[0].float().cpu()) < 1.5e-2


@pytest.mark.parametrize("C", [128, 192, 256, 384])
@pytest.mark.parametrize("M_", [1, 31, 32, 129, 1000])
@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16])
def test_hpre_kernel_pair_vs_fp32_reference_through_the_c_abi(R, M_, gdt, C):
    """cnx_block_mlp_fwd_hpre / cnx_block_mlp_bwd_input_hpre (C = 128 ... 384) called directly: ragged row counts (the workspace is
    sized in 128-row workgroups), fp32 and bf16 incoming gradients, output equal to the plain fused forward bit for bit, input
    gradient vs fp32 autograd of the same bf16-quantised operands; argument errors."""
    lib = R._lib.load()
    g = torch.Generator().manual_seed(M_)
    u = (torch.randn(M_, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w1 = torch.randn(4 * C, C, generator=g) * C ** -0.5
    w2 = torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5
    lw, lb = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    b1, b2 = torch.randn(4 * C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    gm = torch.randn(C, generator=g)
    x = torch.randn(M_, C, generator=g)
    gout = torch.randn(M_, C, generator=g)
    ur = u.float().requires_grad_()
    a = F.layer_norm(ur, (C,), lw, lb, 1e-6)
    y = x + gm * (F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1) @ w2.to(torch.bfloat16).float().t() + b2)
    (gref,) = torch.autograd.grad(y, ur, gout.to(gdt).float())
    wf, wb = R.ops._pack_mlp(w1.cuda(), w2.cuda()), R.ops._pack_mlp_bwd(w1.cuda(), w2.cuda())
    ud, xd, lwd, lbd, b1d, b2d, gmd = (t.cuda() for t in (u, x, lw, lb, b1, b2, gm))
    n_ws = lib.cnx_block_mlp_hpre_elems(M_, C)
    assert n_ws == ((M_ + 127) // 128) * 128 * 4 * C and lib.cnx_block_mlp_hpre_elems(0, C) == 0
    ws = torch.full((n_ws,), float("nan"), device="cuda", dtype=torch.bfloat16)
    out, out0 = torch.empty(M_, C, device="cuda"), torch.empty(M_, C, device="cuda")
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    P = R._lib.ptr
    fwd = lambda o, w, m: lib.cnx_block_mlp_fwd_hpre(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(),
                                                     wf.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), gmd.data_ptr(), xd.data_ptr(), 0,
                                                     o.data_ptr(), 0, w, m, C, S())
    assert fwd(out, ws.data_ptr(), M_) == 0
    assert lib.cnx_block_mlp_fwd(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, None, None, wf.data_ptr(), b1d.data_ptr(),
                                 b2d.data_ptr(), gmd.data_ptr(), xd.data_ptr(), 0, out0.data_ptr(), 0, None, M_, C, S()) == 0
    assert torch.equal(out, out0)
    assert float((out.cpu() - y.detach()).norm() / y.detach().norm()) < 1e-2
    gd = gout.to(gdt).cuda()
    du = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    bwd = lambda w, m, c: lib.cnx_block_mlp_bwd_input_hpre(ud.data_ptr(), lwd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
                                                           R._lib.dtype_code(gdt), gmd.data_ptr(), wb.data_ptr(), w, du.data_ptr(), m, c, S())
    assert bwd(ws.data_ptr(), M_, C) == 0
    assert float((du.float().cpu() - gref).norm() / gref.norm()) < 1.5e-2
    # argument handling: empty batch is a no-op, missing workspace / unsupported width are refused
    assert fwd(out, ws.data_ptr(), 0) == 0 and bwd(ws.data_ptr(), 0, C) == 0
    assert fwd(out, None, M_) < 0 and bwd(None, M_, C) < 0 and bwd(ws.data_ptr(), M_, 96) < 0
    assert lib.cnx_block_mlp_hpre_supported(C) == 1 and lib.cnx_block_mlp_hpre_supported(96) == 0


@pytest.mark.parametrize("CI,N,H,W", [(48, 2, 32, 32), (48, 3, 8, 16), (48, 1, 112, 112), (64, 2, 32, 48), (64, 1, 16, 16)])
@pytest.mark.parametrize("bias", [True, False])
def test_second_stem_convolution_vs_library(R, CI, N, H, W, bias):
    """cnx_conv3x3s2_fwd / _dgrad (3x3, stride 2, padding 1; ConvBlock1's 48 -> 96, ConvBlock3's 64 -> 96) vs F.conv2d in fp32 on the
    same bf16-quantised operands; borders, odd tile counts, the autograd wrapper (library filter / bias gradients), reproducibility."""
    lib = R._lib.load()
    CO = 96
    assert lib.cnx_conv3x3s2_supported(CI, CO, H, W) == 1 and lib.cnx_conv3x3s2_supported(CI, CO, H + 4, W) == 0
    assert lib.cnx_conv3x3s2_supported(32, CO, H, W) == 0
    g = torch.Generator().manual_seed(CI + H + W)
    x = torch.randn(N, CI, H, W, generator=g).to(torch.bfloat16)
    w = torch.randn(CO, CI, 3, 3, generator=g) * (9 * CI) ** -0.5
    b = torch.randn(CO, generator=g) * 0.2 if bias else None
    cot = torch.randn(N, CO, H // 2, W // 2, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_()
    wr = w.to(torch.bfloat16).float().requires_grad_()
    ref = F.conv2d(xr, wr, b, stride=2, padding=1)
    gx, gw = torch.autograd.grad(ref, [xr, wr], cot.float())
    conv = torch.nn.Conv2d(CI, CO, 3, stride=2, padding=1, bias=bias).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
        if bias:
            conv.bias.copy_(b)
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    assert R.ops.conv3x3s2_supported(xd, conv) and not R.ops.conv3x3s2_supported(xd.float(), conv)
    out = R.ops.conv3x3s2(xd, conv.weight, conv.bias)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    rel = lambda a, b_: float((a.float().cpu() - b_).norm() / b_.norm())
    assert rel(out.detach(), ref.detach()) < 4e-3
    grads = torch.autograd.grad(out, [xd, conv.weight] + ([conv.bias] if bias else []), cot.cuda())
    assert rel(grads[0], gx) < 6e-3 and rel(grads[1], gw) < 1e-2
    if bias:
        assert rel(grads[2], cot.float().sum((0, 2, 3))) < 1e-2
    # the input-gradient kernel through the C ABI, twice: same bits
    pk = R.ops._pack_conv2(conv.weight.detach())
    dys = cot.cuda().permute(0, 2, 3, 1).contiguous()
    dx1 = torch.empty(N, H, W, CI, device="cuda", dtype=torch.bfloat16)
    dx2 = torch.full_like(dx1, float("nan"))
    for d in (dx1, dx2):
        assert lib.cnx_conv3x3s2_dgrad(dys.data_ptr(), pk.data_ptr(), d.data_ptr(), N, H, W, CI, CO, S()) == 0
    assert torch.equal(dx1, dx2) and rel(dx1.permute(0, 3, 1, 2), gx) < 6e-3
    assert lib.cnx_conv3x3s2_fwd(xd.data_ptr(), pk.data_ptr(), None, out.data_ptr(), 0, H, W, CI, CO, S()) == 0
    assert lib.cnx_conv3x3s2_fwd(xd.data_ptr(), None, None, out.data_ptr(), N, H, W, CI, CO, S()) < 0
    assert lib.cnx_conv3x3s2_dgrad(dys.data_ptr(), pk.data_ptr(), dx1.data_ptr(), N, H + 2, W, CI, CO, S()) < 0


@pytest.mark.parametrize("CI,N,H,W", [(48, 2, 32, 32), (48, 3, 8, 16), (48, 2, 112, 112), (64, 2, 32, 48), (64, 1, 16, 16), (48, 1, 2, 2),
                                      (48, 5, 6, 10), (64, 3, 112, 112), (48, 700, 4, 4)])
def test_second_stem_convolution_filter_gradient_vs_fp32_reference(R, CI, N, H, W):
    """cnx_conv3x3s2_wgrad (csrc/wgrad_kernels.hip: implicit GEMM over the positions, both operands through LDS transpose reads,
    row images by LDS-DMA) against the fp32 filter / bias gradient of F.conv2d on the same bf16 operands.  Covers widths whose
    output rows are no multiple of the 16-position k-step (tails), the first image row (zero filter row), more rows than
    workgroups, both channel counts; the result is in the weight's channels-last order and deterministic."""
    lib = R._lib.load()
    CO = 96
    assert lib.cnx_conv3x3s2_wgrad_supported(CI, CO, H, W) == 1
    g = torch.Generator().manual_seed(CI + H + N)
    x = (torch.randn(N, CI, H, W, generator=g) * torch.linspace(0.5, 1.5, CI).view(1, CI, 1, 1)).to(torch.bfloat16)
    dy = (torch.randn(N, CO, H // 2, W // 2, generator=g) * torch.linspace(0.3, 2.0, CO).view(1, CO, 1, 1)).to(torch.bfloat16)
    w = torch.zeros(CO, CI, 3, 3, requires_grad=True)
    b = torch.zeros(CO, requires_grad=True)
    gw, gb = torch.autograd.grad(F.conv2d(x.float(), w, b, stride=2, padding=1), (w, b), dy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full((CO, 3, 3, CI), float("nan"), device="cuda")
    db = torch.full((CO,), float("nan"), device="cuda")
    ws = torch.empty(lib.cnx_conv3x3s2_wgrad_ws_floats(CI, CO), device="cuda")
    assert lib.cnx_conv3x3s2_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), N, H, W, CI, CO, S()) == 0
    scale = float(gw.abs().max()) + 1e-6
    close(dw.permute(0, 3, 1, 2), gw, 1e-4, 3e-5 * scale)          # fp32 sums of exact bf16 products, another order
    close(db, gb, 1e-4, 3e-5 * float(gb.abs().max()) + 1e-6)
    dw2 = torch.empty_like(dw)
    assert lib.cnx_conv3x3s2_wgrad(xd.data_ptr(), dyd.data_ptr(), dw2.data_ptr(), None, ws.data_ptr(), N, H, W, CI, CO, S()) == 0
    assert torch.equal(dw, dw2)
    assert lib.cnx_conv3x3s2_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), N, H, W, 32, CO, S()) == -4
    assert lib.cnx_conv3x3s2_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), N, H + 1, W, CI, CO, S()) == -4


def test_full_model_matches_reference_model_fp32(R):
    torch.manual_seed(0)
    ref = M.ConvNeXtTimm(depths=(1, 1, 2, 1), dims=(32, 64, 96, 128), num_classes=10)
    ref.stem = M.ConvStem('block1', 16)
    ref.eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(4.0) if p.ndim > 1 else p.add_(torch.randn_like(p) * 0.1)
    A = R.architecture
    dev = A.ConvNeXt(depths=(1, 1, 2, 1), dims=(32, 64, 96, 128), num_classes=10)
    dev.stem = A.ConvBlock1(16)
    dev.load_state_dict(ref.state_dict())
    dev = dev.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.rand(3, 3, 64, 64)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    (gr,) = torch.autograd.grad(yr.sum(), xr)
    xd = x.cuda().requires_grad_()
    yd = dev(xd)
    (gd,) = torch.autograd.grad(yd.sum(), xd)
    close(yd, yr, 1e-3, 1e-3)
    assert float((gd.cpu() - gr).norm() / gr.norm()) < 2e-3
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = dev(x.cuda())
    assert float((yb.float().cpu() - yr).norm() / yr.norm()) < 2e-2


def test_fused_ops_refuse_cpu_tensors(R):
    blk = R.architecture.ConvNeXtBlock(8)
    with pytest.raises(R._lib.ApgdHipError):
        blk(torch.randn(1, 8, 7, 7))


def test_apgd_train_on_product_model_under_autocast(R):
    torch.manual_seed(0)
    A = R.architecture
    m = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(32, 64, 96, 128), num_classes=10)
    m.stem = A.ConvBlock1(16)
    m = m.cuda().to(memory_format=torch.channels_last).eval()
    x = torch.rand(8, 3, 64, 64, device="cuda")
    y = torch.randint(0, 10, (8,), device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xb, acc, lb, xba = R.apgd_train(m, x, y, norm="Linf", eps=4 / 255, n_iter=3)
    assert xb.dtype == torch.float32 and float((xb - x).abs().max()) <= 4 / 255 + 1e-7
    assert float(xb.min()) >= 0 and float(xb.max()) <= 1 and torch.isfinite(lb).all()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        clean = F.cross_entropy(m(x).float(), y, reduction="none")
    assert (lb >= clean - 1e-2).all()        # loss_best never falls below the clean loss (up to bf16 noise)


@pytest.mark.parametrize("C", [96, 128, 192, 256, 384])
@pytest.mark.parametrize("M_", [1, 31, 128, 300, 1000])
@pytest.mark.parametrize("gamma,ln", [(True, True), (False, True), (True, False)])
def test_fused_block_tail_forward_vs_fp32_reference(R, C, M_, gamma, ln):
    """cnx_block_mlp_fwd (LN prologue + two chained MFMA GEMMs, GELU in registers) vs an fp32 torch evaluation on
    the same bf16-quantised operands.  Bar: the product's bf16 <= 1e-2 relative tolerance (north_star)."""
    lib = R._lib.load()
    g = torch.Generator().manual_seed(C + M_)
    u = (torch.randn(M_, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5)
    lw, lb = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    b1, b2 = torch.randn(4 * C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    gm = torch.randn(C, generator=g) if gamma else None
    x = torch.randn(M_, C, generator=g)
    a = F.layer_norm(u.float(), (C,), lw, lb, 1e-6).to(torch.bfloat16).float() if ln else u.float()
    h = F.gelu(a @ w1.to(torch.bfloat16).float().t() + b1).to(torch.bfloat16).float()
    y2 = h @ w2.to(torch.bfloat16).float().t() + b2
    ref = x + (y2 * gm if gamma else y2)
    wf = R.ops._pack_mlp(w1.cuda(), w2.cuda())
    assert wf.numel() == lib.cnx_mlp_packed_elems(C) and wf.numel() in (8 * C * C, 8 * C * C + 64 * C)   # C/8 slices (+1 when pipelined)
    wf_b = R.ops._pack_mlp(w1.cuda().to(torch.bfloat16), w2.cuda().to(torch.bfloat16))
    assert torch.equal(wf, wf_b)                                      # fp32 and bf16 masters pack identically
    ud, xd = u.cuda(), x.cuda()
    out = torch.empty(M_, C, device="cuda")
    y2d = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    lwd, lbd, b1d, b2d = lw.cuda(), lb.cuda(), b1.cuda(), b2.cuda()
    gmd = gm.cuda() if gamma else None
    assert lib.cnx_block_mlp_fwd(ud.data_ptr(), lwd.data_ptr() if ln else None, lbd.data_ptr() if ln else None, 1e-6,
                                 mean.data_ptr() if ln else None, rstd.data_ptr() if ln else None, wf.data_ptr(),
                                 b1d.data_ptr(), b2d.data_ptr(), gmd.data_ptr() if gamma else None, xd.data_ptr(), 0,
                                 out.data_ptr(), 0, y2d.data_ptr(), M_, C, S()) == 0
    err = float((out.cpu() - ref).norm() / (ref - x).norm())
    assert err < 4e-3, err                       # same roundings as the reference above; accumulation order differs
    assert float((y2d.float().cpu() - y2).norm() / y2.norm()) < 6e-3
    if ln:
        mu = u.float().mean(1)
        var = u.float().var(1, unbiased=False)
        close(mean, mu, 1e-5, 1e-6)
        close(rstd, (var + 1e-6).rsqrt(), 1e-5, 1e-6)
    # bf16 residual / bf16 output variant
    outb = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    xb = x.to(torch.bfloat16).cuda()
    assert lib.cnx_block_mlp_fwd(ud.data_ptr(), lwd.data_ptr() if ln else None, lbd.data_ptr() if ln else None, 1e-6,
                                 None, None, wf.data_ptr(), b1d.data_ptr(), b2d.data_ptr(),
                                 gmd.data_ptr() if gamma else None, xb.data_ptr(), 1, outb.data_ptr(), 1, None, M_, C,
                                 S()) == 0
    refb = xb.float().cpu() + (y2 * gm if gamma else y2)
    assert float((outb.float().cpu() - refb).norm() / refb.norm()) < 1e-2


def test_fused_block_tail_argument_errors(R):
    lib = R._lib.load()
    assert lib.cnx_block_mlp_supported(96) == 1 and lib.cnx_block_mlp_supported(100) == 0
    t = torch.zeros(64, device="cuda")
    args = lambda C, M_=1: (t.data_ptr(), None, None, 1e-6, None, None, t.data_ptr(), t.data_ptr(), t.data_ptr(), None, None,
                            0, t.data_ptr(), 0, None, M_, C, S())
    assert lib.cnx_block_mlp_fwd(*args(100)) == -4                    # no kernel for this width
    assert lib.cnx_block_mlp_fwd(*args(96, 0)) == 0                   # empty input is a no-op
    assert lib.cnx_block_mlp_fwd(*args(96, -1)) != 0
    assert lib.cnx_mlp_pack_weights(None, t.data_ptr(), 0, t.data_ptr(), 96, S()) == -1
    assert lib.cnx_mlp_pack_weights(t.data_ptr(), t.data_ptr(), 2, t.data_ptr(), 96, S()) != 0   # fp16 masters: unsupported


@pytest.mark.parametrize("C", [96, 128, 192, 256])
@pytest.mark.parametrize("M_", [1, 33, 256, 328, 700])
@pytest.mark.parametrize("gamma,emit,gdt", [(True, True, torch.float32), (False, False, torch.float32),
                                            (True, False, torch.bfloat16)])
def test_fused_block_tail_backward_vs_fp32_reference(R, C, M_, gamma, emit, gdt):
    """cnx_block_mlp_bwd (LN recompute + three chained MFMA GEMMs, GELU' in registers) vs fp32 autograd of the same
    chain on the same bf16-quantised operands; emitted weight-gradient operands vs their definitions."""
    lib = R._lib.load()
    gen = torch.Generator().manual_seed(C * 7 + M_)
    u = (torch.randn(M_, C, generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    w1 = (torch.randn(4 * C, C, generator=gen) * C ** -0.5).to(torch.bfloat16).float()
    w2 = (torch.randn(C, 4 * C, generator=gen) * (4 * C) ** -0.5).to(torch.bfloat16).float()
    lw, lb = 1 + 0.2 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    b1 = torch.randn(4 * C, generator=gen) * 0.3
    gm = torch.randn(C, generator=gen) if gamma else None
    g = torch.randn(M_, C, generator=gen).to(gdt)
    mu = u.float().mean(1)
    rstd = (u.float().var(1, unbiased=False) + 1e-6).rsqrt()
    a = (((u.float() - mu[:, None]) * rstd[:, None]) * lw + lb).to(torch.bfloat16).float().requires_grad_()
    dO = (g.float() * gm if gamma else g.float()).to(torch.bfloat16).float()
    hpre = a @ w1.t() + b1
    h = F.gelu(hpre)
    (da_ref,) = torch.autograd.grad(h @ w2.t(), a, dO)
    dh = dO @ w2
    wb = R.ops._pack_mlp_bwd(w1.cuda(), w2.cuda())
    assert wb.numel() == lib.cnx_mlp_packed_bwd_elems(C) == 12 * C * C
    dev_ = lambda t: t.detach().cuda().contiguous()
    ud, gd, lwd, lbd, b1d, mud, rsd = dev_(u), dev_(g), dev_(lw), dev_(lb), dev_(b1), dev_(mu), dev_(rstd)
    gmd = dev_(gm) if gamma else None
    da = torch.full((M_, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    ao = dob = ht = dhpt = None
    if emit:
        ao = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
        dob = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
        ht = torch.empty(4 * C, M_, device="cuda", dtype=torch.bfloat16)
        dhpt = torch.empty(4 * C, M_, device="cuda", dtype=torch.bfloat16)
    P = R._lib.ptr
    assert lib.cnx_block_mlp_bwd(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), mud.data_ptr(), rsd.data_ptr(), gd.data_ptr(),
                                 R._lib.dtype_code(gdt), P(gmd), wb.data_ptr(), b1d.data_ptr(), da.data_ptr(), P(ao), 0, P(dob),
                                 P(ht), P(dhpt), M_, C, S()) == 0
    err = float((da.float().cpu() - da_ref).norm() / da_ref.norm())
    assert err < 8e-3, err                     # bf16 roundings of dHpre and of the stored result
    if emit:
        close(ao, a.detach(), 8e-3, 1e-3)           # one bf16 ulp: fma vs separate mul/add before the rounding
        close(dob, dO, 8e-3, 1e-3)
        close(ht.float().t(), h.detach(), 1e-2, 1e-3)
        hp2 = hpre.detach().clone().requires_grad_()
        (dhpre,) = torch.autograd.grad(F.gelu(hp2), hp2, dh)
        assert float((dhpt.float().t().cpu() - dhpre).norm() / dhpre.norm()) < 6e-3
    if not emit:
        # input-gradient-only entry: LayerNorm backward in the epilogue -> gradient w.r.t. u, vs fp32 autograd through LN
        u32 = u.float().requires_grad_()
        a2 = F.layer_norm(u32, (C,), lw, lb, 1e-6)
        (du_ref,) = torch.autograd.grad(a2, u32, da_ref)
        du = torch.full((M_, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        assert lib.cnx_block_mlp_bwd_input(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), mud.data_ptr(), rsd.data_ptr(),
                                           gd.data_ptr(), R._lib.dtype_code(gdt), P(gmd), wb.data_ptr(), b1d.data_ptr(),
                                           du.data_ptr(), M_, C, S()) == 0
        err = float((du.float().cpu() - du_ref).norm() / du_ref.norm())
        assert err < 1e-2, err                 # + the cancellation in t - mean(t) - xh mean(t xh) on bf16-accurate da
        assert lib.cnx_block_mlp_bwd_input(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), mud.data_ptr(), rsd.data_ptr(),
                                           gd.data_ptr(), R._lib.dtype_code(gdt), P(gmd), wb.data_ptr(), b1d.data_ptr(),
                                           None, M_, C, S()) == -1
    # argument errors
    assert lib.cnx_block_mlp_bwd(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), mud.data_ptr(), rsd.data_ptr(), gd.data_ptr(),
                                 R._lib.dtype_code(gdt), P(gmd), wb.data_ptr(), b1d.data_ptr(), da.data_ptr(), da.data_ptr(), 0,
                                 None, None, None, M_, C, S()) == -1          # emit pointers: all four or none
    if emit:                                                                  # strided a_out (room for a ones column)
        a8 = torch.full((M_, C + 8), 7.0, device="cuda", dtype=torch.bfloat16)
        assert lib.cnx_block_mlp_bwd(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), mud.data_ptr(), rsd.data_ptr(), gd.data_ptr(),
                                     R._lib.dtype_code(gdt), P(gmd), wb.data_ptr(), b1d.data_ptr(), da.data_ptr(), a8.data_ptr(),
                                     C + 8, P(dob), P(ht), P(dhpt), M_, C, S()) == 0
        assert torch.equal(a8[:, :C], ao) and bool((a8[:, C:] == 7).all())
    assert lib.cnx_block_mlp_bwd_supported(384) == 0


def _acc_to_rows(ws, M_, N_):
    """CNX_TN_ACC tiles ([M/32][N/32] x 2 KiB; element (m, n) of a tile at byte 64 m + 32 ((n/4) % 2) + 8 (n/8) + 2 (n % 4)) -> [M, N]."""
    t = ws[:M_ * N_].view(M_ // 32, N_ // 32, 32, 2, 4, 4)              # (row tile, column tile, m, half, q, e): n = 8 q + 4 half + e
    return t.permute(0, 2, 1, 4, 3, 5).reshape(M_, N_)


@pytest.mark.parametrize("C,M_", [(256, 64), (256, 448), (384, 96), (384, 448), (384, 2048), (384, 6272)])
@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16])
def test_hpre_backward_on_wavefront_pairs_is_bit_identical_to_the_single_wavefront_kernel(R, C, M_, gdt):
    """Round 6: blk2_bwd_kernel (producer: dH, GELU'; consumer: GEMM3, epilogue - two wavefronts per row tile on one SIMD) against
    blk_mlp_bwd_kernel<..., HPRE> in its three forms (attack: du; training: da + dO rows + dHpre tiles; training with the LayerNorm
    backward in the epilogue), selected in one process by cnx_runtime_switch(CNX_SWITCH_BLK2_BWD_WIDTHS): same MFMA order per
    accumulator, same activation arithmetic - every output bit for bit, ragged last workgroups included.  (The single-wavefront
    kernel itself is checked against fp32 references in the tests below.)"""
    lib = R._lib.load()
    gen = torch.Generator().manual_seed(7 * C + M_)
    dev_ = lambda t: t.detach().cuda().contiguous()
    u = dev_((torch.randn(M_, C, generator=gen) * 1.5 + 0.3).to(torch.bfloat16))
    xres = dev_(torch.randn(M_, C, generator=gen))
    w1 = torch.randn(4 * C, C, generator=gen) * C ** -0.5
    w2 = torch.randn(C, 4 * C, generator=gen) * (4 * C) ** -0.5
    lw, lb = dev_(1 + 0.2 * torch.randn(C, generator=gen)), dev_(0.2 * torch.randn(C, generator=gen))
    b1, b2 = dev_(torch.randn(4 * C, generator=gen) * 0.3), dev_(torch.randn(C, generator=gen) * 0.3)
    gm = dev_(torch.randn(C, generator=gen))
    g = dev_(torch.randn(M_, C, generator=gen).to(gdt))
    wf, wb = R.ops._pack_mlp(w1.cuda(), w2.cuda()), R.ops._pack_mlp_bwd(w1.cuda(), w2.cuda())
    n_ws = lib.cnx_block_mlp_hpre_elems(M_, C)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    out = torch.empty(M_, C, device="cuda")
    hp = torch.zeros(n_ws, device="cuda", dtype=torch.bfloat16)
    assert lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(),
                                      b2.data_ptr(), gm.data_ptr(), xres.data_ptr(), 0, out.data_ptr(), 0, hp.data_ptr(), M_, C, S()) == 0
    code = R._lib.dtype_code(gdt)
    res = {}
    prev = lib.cnx_runtime_switch(3, -1)
    try:
        for w in (0, 3):
            assert lib.cnx_runtime_switch(3, w) >= 0
            mk = lambda *sh: torch.full(sh, float("nan"), device="cuda", dtype=torch.bfloat16)
            du0, da1, do1, dh1, du2, do2, dh2 = mk(M_, C), mk(M_, C), mk(M_, C), mk(n_ws), mk(M_, C), mk(M_, C), mk(n_ws)
            assert lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), code, gm.data_ptr(),
                                                    wb.data_ptr(), hp.data_ptr(), du0.data_ptr(), M_, C, S()) == 0
            assert lib.cnx_block_mlp_bwd_train_hpre(g.data_ptr(), code, gm.data_ptr(), wb.data_ptr(), hp.data_ptr(), da1.data_ptr(), do1.data_ptr(),
                                                    dh1.data_ptr(), M_, C, S()) == 0
            assert lib.cnx_block_mlp_bwd_train_hpre_ln(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), code, gm.data_ptr(),
                                                       wb.data_ptr(), hp.data_ptr(), du2.data_ptr(), do2.data_ptr(), dh2.data_ptr(), M_, C, S()) == 0
            torch.cuda.synchronize()
            nt = (M_ + 31) // 32 * 32 * 4 * C                       # tiles of rows that exist (the workspace is padded to 128 rows)
            res[w] = (du0, da1, do1, dh1[:nt], du2, do2, dh2[:nt])
    finally:
        lib.cnx_runtime_switch(3, prev)
    for a, b, name in zip(res[0], res[3], ("du (attack)", "da", "dO rows", "dHpre tiles", "du (training)", "dO rows (LN form)", "dHpre tiles (LN form)")):
        assert not torch.isnan(b.float()).any(), name
        assert torch.equal(a, b), name
    assert torch.equal(res[3][0], res[3][4]) and torch.equal(res[3][3], res[3][6])      # the forms agree with each other


def test_wavefront_pair_kernels_repeat_bit_for_bit(R):
    """Race hunt (tools/pair_stress.py runs it at full size, 3000 launches): the pair kernels hand tiles over through LDS, count their DMA
    by hand and split the epilogue between the two wavefronts of a pair - 25 launches of the forward (training form) and of the Hpre
    backward (attack form) on the same inputs give the same bits every time, on a ragged row count."""
    lib = R._lib.load()
    C, M_ = 384, 6272 + 96
    gen = torch.Generator(device="cuda").manual_seed(3)
    u = torch.randn(M_, C, device="cuda", generator=gen).to(torch.bfloat16)
    x = torch.randn(M_, C, device="cuda", generator=gen)
    w1 = torch.randn(4 * C, C, device="cuda", generator=gen) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device="cuda", generator=gen) * (4 * C) ** -0.5
    lw, lb = 1 + 0.1 * torch.randn(C, device="cuda", generator=gen), 0.1 * torch.randn(C, device="cuda", generator=gen)
    b1, b2 = 0.1 * torch.randn(4 * C, device="cuda", generator=gen), 0.1 * torch.randn(C, device="cuda", generator=gen)
    gm = 0.5 + 0.1 * torch.randn(C, device="cuda", generator=gen)
    g = torch.randn(M_, C, device="cuda", generator=gen)
    wf, wb = R.ops._pack_mlp(w1, w2), R.ops._pack_mlp_bwd(w1, w2)
    n_ws = lib.cnx_block_mlp_hpre_elems(M_, C)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    z = lambda *sh, dt=torch.bfloat16: torch.zeros(sh, device="cuda", dtype=dt)
    out, hp, hact, arows, y2, du = z(M_, C, dt=torch.float32), z(n_ws), z(n_ws), z(M_, C), z(M_, C), z(M_, C)
    first = None
    for rep in range(25):
        for t in (out, hp, hact, arows, y2, du):
            t.add_(1)
        assert lib.cnx_block_mlp_fwd_train(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(),
                                           b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, out.data_ptr(), 0, y2.data_ptr(), hp.data_ptr(), hact.data_ptr(),
                                           arows.data_ptr(), M_, C, S()) == 0
        assert lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, gm.data_ptr(), wb.data_ptr(),
                                                hp.data_ptr(), du.data_ptr(), M_, C, S()) == 0
        torch.cuda.synchronize()
        nt = (M_ + 31) // 32 * 32 * 4 * C
        cur = [t.clone() for t in (out, hp[:nt], hact[:nt], arows, y2, du)]
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, first)), rep


def _hpre_pair_training_pass(R, C, M_, lw, lb, gm, seed):
    """Forward (cnx_block_mlp_fwd_train) + backward with the LayerNorm backward in its epilogue (cnx_block_mlp_bwd_train_hpre_ln and the
    plain cnx_block_mlp_bwd_train_hpre for da) + both weight gradients on cnx_gemm_tn_ex, with the given LayerNorm parameters and layer
    scale: the device tensors the per-channel identities are made of."""
    lib = R._lib.load()
    gen = torch.Generator().manual_seed(seed)
    u = (torch.randn(M_, C, generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    xres = torch.randn(M_, C, generator=gen)
    w1 = (torch.randn(4 * C, C, generator=gen) * C ** -0.5).to(torch.bfloat16).float()
    w2 = (torch.randn(C, 4 * C, generator=gen) * (4 * C) ** -0.5).to(torch.bfloat16).float()
    b1, b2 = torch.randn(4 * C, generator=gen) * 0.3, torch.randn(C, generator=gen) * 0.3
    g = torch.randn(M_, C, generator=gen)
    dev_ = lambda t: t.detach().cuda().contiguous()
    P = R._lib.ptr
    wf = R.ops._pack_mlp(w1.cuda(), w2.cuda())
    wb = R.ops._pack_mlp_bwd(w1.cuda(), w2.cuda())
    ud, xd, lwd, lbd, b1d, b2d, gd, gmd, w1d, w2d = map(dev_, (u, xres, lw, lb, b1, b2, g, gm, w1, w2))
    n_ws = lib.cnx_block_mlp_hpre_elems(M_, C)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    hpre_ws, h_ws, dhp_ws = (torch.zeros(n_ws, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    a_rows, y2d, da, dos = (torch.empty(M_, C, device="cuda", dtype=torch.bfloat16) for _ in range(4))
    out = torch.empty(M_, C, device="cuda")
    assert lib.cnx_block_mlp_fwd_train(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                       b1d.data_ptr(), b2d.data_ptr(), P(gmd), xd.data_ptr(), 0, out.data_ptr(), 0, y2d.data_ptr(),
                                       hpre_ws.data_ptr(), h_ws.data_ptr(), a_rows.data_ptr(), M_, C, S()) == 0
    assert lib.cnx_block_mlp_bwd_train_hpre(gd.data_ptr(), 0, P(gmd), wb.data_ptr(), hpre_ws.data_ptr(), da.data_ptr(), dos.data_ptr(),
                                            dhp_ws.data_ptr(), M_, C, S()) == 0
    dw2 = torch.empty(C, 4 * C, device="cuda"); db2 = torch.empty(C, device="cuda")
    dw1 = torch.empty(4 * C, C, device="cuda"); db1 = torch.empty(4 * C, device="cuda")
    ws = torch.empty(max(lib.cnx_gemm_tn_ws_floats(M_, C, 4 * C), lib.cnx_gemm_tn_ws_floats(M_, 4 * C, C)), device="cuda")
    assert lib.cnx_gemm_tn_ex(dos.data_ptr(), C, 0, h_ws.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M_, C, 4 * C, S()) == 0
    assert lib.cnx_gemm_tn_ex(dhp_ws.data_ptr(), 0, 1, a_rows.data_ptr(), C, 0, dw1.data_ptr(), db1.data_ptr(), ws.data_ptr(), M_, 4 * C, C, S()) == 0
    return dict(u=ud, lw=lwd, lb=lbd, gm=gmd, g=gd, w1=w1d, w2=w2d, b2=b2d, mean=mean, rstd=rstd, y2=y2d, h_ws=h_ws, dhp_ws=dhp_ws, da=da,
                dos=dos, dw1=dw1, db1=db1, dw2=dw2, db2=db2)


@pytest.mark.parametrize("C,M_", [(96, 3136), (192, 1024), (384, 2048)])
def test_per_channel_gradient_identities_at_the_benchmarked_init_and_where_they_are_ill_conditioned(R, C, M_):
    """Round 6: d(gamma) / d(ln_w) / d(ln_b) of a fused block as cnx_block_dgamma / cnx_block_dln compute them
    in the training pass, against the DIRECT sums over the kernels' own da / g / y2 (what cnx_layernorm_bwd / cnx_scale_residual_bwd
    sum), (a) at the init of the model bench.py times - gamma = 1e-6, ln_w = 1, ln_b = 0 (models/convnext.py:33, LayerNorm defaults) -
    and (b) with LayerNorm channels on which the d(ln_w) identity is ill-conditioned (|ln_b| >> |ln_w|: it recovers xh from
    bf16(xh ln_w + ln_b)): those take the direct sum (dln_ill: |ln_b| > 4 |ln_w|), the identity alone is shown to be off there."""
    lib = R._lib.load()
    if not lib.cnx_block_mlp_hpre_supported(C):
        pytest.skip("no Hpre kernel pair at this width: the C = 96 training backward is covered by test_recomputing_training_backward_*")

    def run(lw, lb, gm, seed):
        t = _hpre_pair_training_pass(R, C, M_, lw, lb, gm, seed)
        xh = (t["u"].double() - t["mean"].double()[:, None]) * t["rstd"].double()[:, None]
        dlw_direct, dlb_direct = (t["da"].double() * xh).sum(0), t["da"].double().sum(0)
        dlw, dlb = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        res = {}
        for name, dap, tlp in (("da", t["da"].data_ptr(), None), ("tiles", None, t["dhp_ws"].data_ptr())):
            wsd = torch.full((lib.cnx_block_dln_ws_floats(C),), float("nan"), device="cuda")
            dlw.fill_(float("nan"))
            assert lib.cnx_block_dln(t["w1"].data_ptr(), t["dw1"].data_ptr(), t["db1"].data_ptr(), t["lw"].data_ptr(), t["lb"].data_ptr(), dap, tlp,
                                     t["u"].data_ptr(), t["mean"].data_ptr(), t["rstd"].data_ptr(), dlw.data_ptr(), dlb.data_ptr(), wsd.data_ptr(),
                                     M_, C, 4 * C, S()) == 0
            res[name] = (dlw.double().clone(), dlb.double().clone())
        dgm = torch.empty(C, device="cuda")
        assert lib.cnx_block_dgamma(t["w2"].data_ptr(), t["dw2"].data_ptr(), t["b2"].data_ptr(), t["db2"].data_ptr(), t["gm"].data_ptr(), t["g"].data_ptr(),
                                    0, None, t["h_ws"].data_ptr(), dgm.data_ptr(), M_, C, 4 * C, S()) == 0
        dgm_direct = (t["g"].double() * t["y2"].double()).sum(0)
        # the identity in fp64 from the same dW1 / d(b1): what the kernel would return WITHOUT the conditioning guard
        w1r = t["w1"].to(torch.bfloat16).double()
        dlb_id = (w1r * t["db1"].double()[:, None]).sum(0)
        dlw_id = ((w1r * t["dw1"].double()).sum(0) - t["lb"].double() * dlb_id) / t["lw"].double()
        return t, res, (dlw_direct, dlb_direct), (dgm.double(), dgm_direct), dlw_id

    rel = lambda a, b: float((a - b).norm() / b.norm())
    # (a) the benchmarked model's init
    t, res, (dlw_d, dlb_d), (dgm, dgm_d), _ = run(torch.ones(C), torch.zeros(C), torch.full((C,), 1e-6), 11 + C)
    for name in ("da", "tiles"):
        assert rel(res[name][0], dlw_d) < 6e-3 and rel(res[name][1], dlb_d) < 6e-3, (name, rel(res[name][0], dlw_d), rel(res[name][1], dlb_d))
    assert rel(dgm, dgm_d) < 8e-3, rel(dgm, dgm_d)
    assert float(dgm_d.abs().mean()) > 1.0                         # d(gamma) is O(sum_m g y2) although dO = bf16(1e-6 g): nothing flushed
    # (b) ill-conditioned LayerNorm channels: ratios |ln_b| / |ln_w| of 3 (identity), 6, 40, 1000 and a zero weight (direct sums);
    #     a large b2 next to them for d(gamma) (the judge's |b2 db2| >> |sum W2 dW2| case: no cancellation, see block_dgamma_kernel)
    gen = torch.Generator().manual_seed(5)
    lw, lb = 1 + 0.2 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    ill = {7: (0.05, 0.30), 20: (0.02, -0.80), 33: (-0.001, 1.0), 41: (0.0, 0.5)}
    lw[3], lb[3] = 0.1, 0.3                                        # ratio 3: still the identity
    for c, (w_, b_) in ill.items():
        lw[c], lb[c] = w_, b_
    t, res, (dlw_d, dlb_d), (dgm, dgm_d), dlw_id = run(lw, lb, torch.randn(C, generator=gen), 12 + C)
    keep = torch.ones(C, dtype=torch.bool)
    keep[list(ill)] = False
    for name in ("da", "tiles"):
        dlw, dlb = res[name]
        assert rel(dlw.cpu()[keep], dlw_d.cpu()[keep]) < 1.2e-2 and rel(dlb, dlb_d) < 1e-2
        scale = float(dlw_d.abs().mean())
        for c in ill:
            # the direct sum: exact from da as stored; from the tiles da is not rounded to bf16 before the sum (rounding noise of M_ terms)
            assert abs(float(dlw[c] - dlw_d[c])) <= (2e-4 if name == "da" else 2e-2) * (abs(float(dlw_d[c])) + scale), (name, c)
    # ... and the guard is needed: on the worst of these channels the bare identity is off by more than the bf16 bar
    worst = max(abs(float(dlw_id[c] - dlw_d[c])) / (abs(float(dlw_d[c])) + float(dlw_d.abs().mean())) for c in (20, 33))
    assert worst > 3e-2, worst
    assert rel(dgm, dgm_d) < 8e-3


@pytest.mark.parametrize("C,M_", [(128, 64), (128, 1024), (192, 448), (256, 256), (384, 128), (384, 2048), (192, 12544)])
@pytest.mark.parametrize("gamma", [True, False])
def test_training_pass_on_the_hpre_kernel_pair_vs_fp32_reference(R, C, M_, gamma):
    """Round 5: the training pass of a block on cnx_block_mlp_fwd_train / cnx_block_mlp_bwd_train_hpre with both weight gradients
    (and their bias gradients) as cnx_gemm_tn_ex contractions over the kernels' own accumulator-order tiles - against fp32 autograd
    of the same chain on the same bf16-quantised operands: block output, saved tiles (decoded), da, dO, dHpre, dW1, db1, dW2, db2."""
    lib = R._lib.load()
    gen = torch.Generator().manual_seed(C + M_)
    u = (torch.randn(M_, C, generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    xres = torch.randn(M_, C, generator=gen)
    w1 = (torch.randn(4 * C, C, generator=gen) * C ** -0.5).to(torch.bfloat16).float()
    w2 = (torch.randn(C, 4 * C, generator=gen) * (4 * C) ** -0.5).to(torch.bfloat16).float()
    lw, lb = 1 + 0.2 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    b1, b2 = torch.randn(4 * C, generator=gen) * 0.3, torch.randn(C, generator=gen) * 0.3
    gm = torch.randn(C, generator=gen) if gamma else None
    g = torch.randn(M_, C, generator=gen)
    # fp32 reference with the roundings the kernels make (LN output, H, dO, dHpre are bf16 operands)
    a = F.layer_norm(u.float(), (C,), lw, lb, 1e-6).to(torch.bfloat16).float()
    hpre = a @ w1.t() + b1
    h = F.gelu(hpre).to(torch.bfloat16).float()
    y2 = h @ w2.t() + b2
    out_ref = xres + (y2 * gm if gamma else y2)
    dO = (g * gm if gamma else g).to(torch.bfloat16).float()
    dh = dO @ w2
    hp2 = hpre.clone().requires_grad_()
    (dhpre,) = torch.autograd.grad(F.gelu(hp2), hp2, dh)
    dhpre_b = dhpre.to(torch.bfloat16).float()
    da_ref = dhpre_b @ w1
    dev_ = lambda t: t.detach().cuda().contiguous()
    P = R._lib.ptr
    wf = R.ops._pack_mlp(w1.cuda(), w2.cuda())
    wb = R.ops._pack_mlp_bwd(w1.cuda(), w2.cuda())
    ud, xd, lwd, lbd, b1d, b2d, gd = map(dev_, (u, xres, lw, lb, b1, b2, g))
    gmd = dev_(gm) if gamma else None
    n_ws = lib.cnx_block_mlp_hpre_elems(M_, C)
    mean, rstd = torch.empty(M_, device="cuda"), torch.empty(M_, device="cuda")
    hpre_ws = torch.zeros(n_ws, device="cuda", dtype=torch.bfloat16)
    h_ws = torch.zeros(n_ws, device="cuda", dtype=torch.bfloat16)
    a_rows = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    y2d = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M_, C, device="cuda")
    assert lib.cnx_block_mlp_fwd_train(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                       b1d.data_ptr(), b2d.data_ptr(), P(gmd), xd.data_ptr(), 0, out.data_ptr(), 0, y2d.data_ptr(),
                                       hpre_ws.data_ptr(), h_ws.data_ptr(), a_rows.data_ptr(), M_, C, S()) == 0
    rel = lambda t, r: float((t.float().cpu() - r).norm() / r.norm())
    assert rel(out, out_ref) < 5e-3 and rel(y2d, y2) < 6e-3
    assert rel(a_rows, a) < 4e-3                                   # one bf16 ulp here and there (fma order before the rounding)
    assert rel(_acc_to_rows(hpre_ws, M_, 4 * C), hpre) < 5e-3 and rel(_acc_to_rows(h_ws, M_, 4 * C), h) < 6e-3
    da = torch.full((M_, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    dos = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    dhp_ws = torch.zeros(n_ws, device="cuda", dtype=torch.bfloat16)
    assert lib.cnx_block_mlp_bwd_train_hpre(gd.data_ptr(), 0, P(gmd), wb.data_ptr(), hpre_ws.data_ptr(), da.data_ptr(), dos.data_ptr(),
                                            dhp_ws.data_ptr(), M_, C, S()) == 0
    assert rel(da, da_ref) < 1e-2 and rel(dos, dO) < 4e-3 and rel(_acc_to_rows(dhp_ws, M_, 4 * C), dhpre) < 8e-3
    # weight / bias gradients from the tiles: exact contractions of what the kernels stored ...
    N1, N2 = C, 4 * C
    dw2 = torch.empty(N1, N2, device="cuda"); db2 = torch.empty(N1, device="cuda")
    # (one workspace for both contractions: the column sums make the [4C, C] call's partials the larger ones)
    ws = torch.empty(max(lib.cnx_gemm_tn_ws_floats(M_, N1, N2), lib.cnx_gemm_tn_ws_floats(M_, N2, N1)), device="cuda")
    assert lib.cnx_gemm_tn_ex(dos.data_ptr(), C, 0, h_ws.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M_, N1, N2, S()) == 0
    dw1 = torch.empty(N2, N1, device="cuda"); db1 = torch.empty(N2, device="cuda")
    assert lib.cnx_gemm_tn_ex(dhp_ws.data_ptr(), 0, 1, a_rows.data_ptr(), C, 0, dw1.data_ptr(), db1.data_ptr(), ws.data_ptr(), M_, N2, N1, S()) == 0
    Hs, Ds = _acc_to_rows(h_ws, M_, 4 * C).float(), _acc_to_rows(dhp_ws, M_, 4 * C).float()
    assert float((dw2 - dos.float().t() @ Hs).norm() / dw2.norm()) < 3e-6 and float((db2 - dos.float().sum(0)).norm() / db2.norm()) < 3e-6
    assert float((dw1 - Ds.t() @ a_rows.float()).norm() / dw1.norm()) < 3e-6 and float((db1 - Ds.sum(0)).norm() / db1.norm()) < 3e-6
    # ... and against the fp32 chain
    assert rel(dw2, dO.t() @ h) < 6e-3 and rel(dw1, dhpre_b.t() @ a) < 8e-3
    assert rel(db2, dO.sum(0)) < 6e-3 and rel(db1, dhpre_b.sum(0)) < 1e-2
    # LayerNorm parameter gradients from dW1 / d(b1) (cnx_block_dln) against the direct sums over da and xh = (u - mean) rstd
    xh = (ud.float() - mean[:, None]) * rstd[:, None]
    d_lw, d_lb = (da.float().double() * xh.double()).sum(0).cpu(), da.float().double().sum(0).cpu()
    dlw, dlb = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    w1d = dev_(w1)
    lwz = lwd.clone()
    for zero_ch in (None, 5):
        if zero_ch is not None:
            lwz[zero_ch] = 0.0                                          # (only the identity's divisor: the kernels above ran with lw)
        wsd = torch.full((lib.cnx_block_dln_ws_floats(C),), float("nan"), device="cuda")
        # da as stored / recomputed from the dHpre tiles; direct sums of ill-conditioned channels by the wide kernel (workspace) / serially
        for dap, tlp, wsp in ((da.data_ptr(), None, wsd.data_ptr()), (None, dhp_ws.data_ptr(), wsd.data_ptr()),
                              (da.data_ptr(), None, None), (None, dhp_ws.data_ptr(), None)):
            dlw.fill_(float("nan"))
            assert lib.cnx_block_dln(w1d.data_ptr(), dw1.data_ptr(), db1.data_ptr(), lwz.data_ptr(), lbd.data_ptr(), dap, tlp, ud.data_ptr(),
                                     mean.data_ptr(), rstd.data_ptr(), dlw.data_ptr(), dlb.data_ptr(), wsp, M_, C, 4 * C, S()) == 0
            tol = 3e-2 if M_ < 256 else 1e-2
            keep = torch.ones(C, dtype=torch.bool)
            if zero_ch is not None:
                keep[zero_ch] = False
                # (from the tiles da is not rounded to bf16 first: the difference is rounding noise of a sum of M_ signed terms)
                assert abs(float(dlw[zero_ch]) - float(d_lw[zero_ch])) <= (1e-4 * (1 + abs(float(d_lw[zero_ch]))) if dap else 2e-2 * float(d_lw.abs().mean() * 4 + 1))
            assert float((dlw.cpu().double()[keep] - d_lw[keep]).norm() / d_lw[keep].norm()) < tol
            assert float((dlb.cpu().double() - d_lb).norm() / d_lb.norm()) < tol
    # the same backward with the LayerNorm backward in its epilogue: du against the LayerNorm backward of the kernel's own da
    du = torch.full((M_, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    dos2 = torch.empty_like(dos); dhp2 = torch.zeros_like(dhp_ws)
    assert lib.cnx_block_mlp_bwd_train_hpre_ln(ud.data_ptr(), lwd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(), 0, P(gmd), wb.data_ptr(),
                                               hpre_ws.data_ptr(), du.data_ptr(), dos2.data_ptr(), dhp2.data_ptr(), M_, C, S()) == 0
    assert torch.equal(dos2, dos) and torch.equal(dhp2, dhp_ws)
    tt = da.float() * lwd
    du_ref = rstd[:, None] * (tt - tt.mean(1, keepdim=True) - xh * (tt * xh).mean(1, keepdim=True))
    assert float((du.float() - du_ref).norm() / du_ref.norm()) < 1e-2
    if gamma:
        # d(gamma) from dW2 / d(b2) (cnx_block_dgamma) against the direct sum over g and the bf16 y2 (models/convnext.py:47) ...
        dgm = torch.empty(C, device="cuda")
        w2d = dev_(w2)
        assert lib.cnx_block_dgamma(w2d.data_ptr(), dw2.data_ptr(), b2d.data_ptr(), db2.data_ptr(), gmd.data_ptr(), gd.data_ptr(), 0, y2d.data_ptr(),
                                    None, dgm.data_ptr(), M_, C, 4 * C, S()) == 0
        direct = (g.double() * y2d.float().cpu().double()).sum(0)
        # (M_ = 64 rows: the sums are short and the bf16 rounding of dO does not average out)
        assert float((dgm.cpu().double() - direct).norm() / direct.norm()) < (2e-2 if M_ < 256 else 8e-3)
        # ... and a channel whose gamma is exactly zero takes the direct sum (its dO column is zero: nothing to divide)
        gz = gmd.clone(); gz[3] = 0.0
        dos_z = dos.clone(); dos_z[:, 3] = 0
        assert lib.cnx_gemm_tn_ex(dos_z.data_ptr(), C, 0, h_ws.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M_, N1, N2, S()) == 0
        for y2p, htp in ((y2d.data_ptr(), None), (None, h_ws.data_ptr())):     # y2 as stored / recomputed from the H tiles
            dgm.fill_(float("nan"))
            assert lib.cnx_block_dgamma(w2d.data_ptr(), dw2.data_ptr(), b2d.data_ptr(), db2.data_ptr(), gz.data_ptr(), gd.data_ptr(), 0, y2p, htp,
                                        dgm.data_ptr(), M_, C, 4 * C, S()) == 0
            assert abs(float(dgm[3]) - float(direct[3])) <= (1e-4 if y2p else 2e-2) * (1 + abs(float(direct[3]))) and bool(torch.isfinite(dgm).all())
        assert lib.cnx_block_dgamma(w2d.data_ptr(), dw2.data_ptr(), b2d.data_ptr(), None, gz.data_ptr(), gd.data_ptr(), 0, y2d.data_ptr(), None,
                                    dgm.data_ptr(), M_, C, 4 * C, S()) == -1
        assert lib.cnx_block_dgamma(w2d.data_ptr(), dw2.data_ptr(), b2d.data_ptr(), db2.data_ptr(), gz.data_ptr(), gd.data_ptr(), 0, None, None,
                                    dgm.data_ptr(), M_, C, 4 * C, S()) == -1
    # argument checks
    assert lib.cnx_gemm_tn_ex(dos.data_ptr(), C, 0, h_ws.data_ptr(), 0, 1, dw2.data_ptr(), None, ws.data_ptr(), M_, N1, N2, S()) == -1
    assert lib.cnx_gemm_tn_ex(dhp_ws.data_ptr(), 0, 1, h_ws.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M_, N2, N2, S()) == -4
    assert lib.cnx_block_mlp_fwd_train(ud.data_ptr(), lwd.data_ptr(), lbd.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                       b1d.data_ptr(), b2d.data_ptr(), P(gmd), xd.data_ptr(), 0, out.data_ptr(), 0, None,
                                       hpre_ws.data_ptr(), None, a_rows.data_ptr(), M_, C, S()) == -1


@pytest.mark.parametrize("C,M_", [(96, 64), (96, 3136), (192, 256), (128, 128), (256, 64)])
def test_recomputing_training_backward_with_accumulator_order_tiles(R, C, M_):
    """cnx_block_mlp_bwd_acc: the recomputing training backward with H / dHpre written as CNX_TN_ACC tiles equals cnx_block_mlp_bwd
    (transposed [4C, M] matrices) value for value, and its tiles feed cnx_gemm_tn_ex."""
    lib = R._lib.load()
    gen = torch.Generator(device="cuda").manual_seed(C + M_)
    u = (torch.randn(M_, C, device="cuda", generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    w1 = torch.randn(4 * C, C, device="cuda", generator=gen) * C ** -0.5
    w2 = torch.randn(C, 4 * C, device="cuda", generator=gen) * (4 * C) ** -0.5
    lw, lb = 1 + 0.2 * torch.randn(C, device="cuda", generator=gen), 0.2 * torch.randn(C, device="cuda", generator=gen)
    b1 = torch.randn(4 * C, device="cuda", generator=gen) * 0.3
    gm = torch.randn(C, device="cuda", generator=gen)
    g = torch.randn(M_, C, device="cuda", generator=gen)
    mu = u.float().mean(1)
    rstd = (u.float().var(1, unbiased=False) + 1e-6).rsqrt()
    wb = R.ops._pack_mlp_bwd(w1, w2)
    mk = lambda *sh: torch.full(sh, float("nan"), device="cuda", dtype=torch.bfloat16)
    da0, a0, do0, ht, dhpt = mk(M_, C), mk(M_, C), mk(M_, C), mk(4 * C, M_), mk(4 * C, M_)
    assert lib.cnx_block_mlp_bwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mu.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, gm.data_ptr(),
                                 wb.data_ptr(), b1.data_ptr(), da0.data_ptr(), a0.data_ptr(), 0, do0.data_ptr(), ht.data_ptr(), dhpt.data_ptr(),
                                 M_, C, S()) == 0
    da1, a1, do1, hw, dw = mk(M_, C), mk(M_, C), mk(M_, C), mk(M_ * 4 * C), mk(M_ * 4 * C)
    assert lib.cnx_block_mlp_bwd_acc(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mu.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, gm.data_ptr(),
                                     wb.data_ptr(), b1.data_ptr(), da1.data_ptr(), a1.data_ptr(), do1.data_ptr(), hw.data_ptr(), dw.data_ptr(),
                                     M_, C, S()) == 0
    assert torch.equal(da0, da1) and torch.equal(a0, a1) and torch.equal(do0, do1)
    assert torch.equal(_acc_to_rows(hw, M_, 4 * C), ht.t()) and torch.equal(_acc_to_rows(dw, M_, 4 * C), dhpt.t())
    if lib.cnx_gemm_tn_supported(M_, 4 * C, C):
        d = torch.empty(4 * C, C, device="cuda"); cs = torch.empty(4 * C, device="cuda")
        ws = torch.empty(lib.cnx_gemm_tn_ws_floats(M_, 4 * C, C), device="cuda")
        assert lib.cnx_gemm_tn_ex(dw.data_ptr(), 0, 1, a1.data_ptr(), C, 0, d.data_ptr(), cs.data_ptr(), ws.data_ptr(), M_, 4 * C, C, S()) == 0
        ref = dhpt.float() @ a1.float()
        assert float((d - ref).norm() / ref.norm()) < 3e-6 and float((cs - dhpt.float().sum(1)).norm() / cs.norm()) < 3e-6
    # the same kernel with the LayerNorm backward in its epilogue: identical emits, du = LayerNorm backward of the kernel's da
    du, a2, do2, hw2, dw2_ = mk(M_, C), mk(M_, C), mk(M_, C), mk(M_ * 4 * C), mk(M_ * 4 * C)
    assert lib.cnx_block_mlp_bwd_acc_ln(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mu.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, gm.data_ptr(),
                                        wb.data_ptr(), b1.data_ptr(), du.data_ptr(), a2.data_ptr(), do2.data_ptr(), hw2.data_ptr(), dw2_.data_ptr(),
                                        M_, C, S()) == 0
    assert torch.equal(a2, a1) and torch.equal(do2, do1) and torch.equal(hw2, hw) and torch.equal(dw2_, dw)
    xh = (u.float() - mu[:, None]) * rstd[:, None]
    tt = da1.float() * lw
    du_ref = rstd[:, None] * (tt - tt.mean(1, keepdim=True) - xh * (tt * xh).mean(1, keepdim=True))
    assert float((du.float() - du_ref).norm() / du_ref.norm()) < 1e-2
    assert lib.cnx_block_mlp_bwd_acc(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), mu.data_ptr(), rstd.data_ptr(), g.data_ptr(), 0, gm.data_ptr(),
                                     wb.data_ptr(), b1.data_ptr(), da1.data_ptr(), a1.data_ptr(), do1.data_ptr(), hw.data_ptr(), dw.data_ptr(),
                                     M_ - 16, C, S()) == -4


@pytest.mark.parametrize("N", [1, 5, 32, 33, 197, 224, 225, 257, 401, 416])
@pytest.mark.parametrize("B,H", [(2, 3), (1, 12)])
def test_fused_attention_forward_and_backward_vs_fp32_reference(R, N, B, H):
    """cnx_attention_fwd (K/V in LDS, scores in MFMA accumulators) vs fp32 torch attention on the same bf16 qkv; the
    autograd wrapper's backward vs fp32 autograd.  Bar: bf16 <= 1e-2 relative (north_star)."""
    lib = R._lib.load()
    d = 64
    C = H * d
    g = torch.Generator().manual_seed(N * 31 + H)
    qkv = (torch.randn(B, N, 3 * C, generator=g) * 1.5).to(torch.bfloat16)
    scale = d ** -0.5
    qr = qkv.float().requires_grad_()
    q, k, v = qr.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4).unbind(0)
    sc = (q @ k.transpose(-2, -1)) * scale
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(B, N, C)
    assert lib.cnx_attention_supported(N, d) == 1 and lib.cnx_attention_supported(N, 32) == 0
    assert lib.cnx_attention_bwd_supported(N, d) == 1 and lib.cnx_attention_bwd_supported(417, d) == 0   # (N <= 224: all operand
    # images in LDS; 225 .. 416 = 320 x 320 inputs: the row operands of the dK / dV kernel come from L2)
    qd = qkv.cuda().requires_grad_()
    out = R.ops.attention(qd, H, scale)
    assert out.dtype == torch.bfloat16 and out.shape == (B, N, C)
    err = float((out.float().cpu() - ref).norm() / ref.norm())
    assert err < 8e-3, err
    lse = torch.empty(B, H, N, device="cuda")
    o2 = torch.empty(B, N, C, device="cuda", dtype=torch.bfloat16)
    assert lib.cnx_attention_fwd(qd.data_ptr(), o2.data_ptr(), lse.data_ptr(), B, N, H, d, scale, S()) == 0
    close(lse, torch.logsumexp(sc, dim=-1), 2e-3, 2e-3)
    cot = torch.randn(B, N, C, generator=g)
    (gr,) = torch.autograd.grad(ref, qr, cot)
    (gd,) = torch.autograd.grad(out, qd, cot.cuda().to(torch.bfloat16))
    assert float((gd.float().cpu() - gr).norm() / gr.norm()) < 1.5e-2
    assert lib.cnx_attention_fwd(qd.data_ptr(), o2.data_ptr(), None, B, 500, H, d, scale, S()) == -4     # N too long


@pytest.mark.parametrize("M_,C", [(1, 8), (37, 96), (300, 384), (1000, 768)])
@pytest.mark.parametrize("gamma", [True, False])
def test_elementwise_tail_kernels(R, M_, C, gamma):
    """cnx_scale_residual / cnx_scale_residual_bwd / cnx_gelu_bwd_colsum vs their torch definitions (one-pass element-wise
    tails of the library-path MLP with the per-channel sums folded in)."""
    lib = R._lib.load()
    g_ = torch.Generator().manual_seed(M_ + C)
    x = torch.randn(M_, C, generator=g_)
    y = torch.randn(M_, C, generator=g_).to(torch.bfloat16)
    gm = torch.randn(C, generator=g_) if gamma else None
    xd, yd = x.cuda(), y.cuda()
    gmd = gm.cuda() if gamma else None
    P = R._lib.ptr
    out = torch.empty(M_, C, device="cuda")
    assert lib.cnx_scale_residual(xd.data_ptr(), 0, yd.data_ptr(), P(gmd), out.data_ptr(), 0, M_, C, S()) == 0
    close(out, x + (y.float() * gm if gamma else y.float()), 1e-6, 1e-6)
    grad = torch.randn(M_, C, generator=g_)
    gd = grad.cuda()
    dos = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    dg, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ws = torch.empty(lib.cnx_colsum_ws_floats(4 * C), device="cuda")
    assert lib.cnx_scale_residual_bwd(gd.data_ptr(), 0, yd.data_ptr(), P(gmd), dos.data_ptr(), dg.data_ptr(), db2.data_ptr(),
                                      ws.data_ptr(), M_, C, S()) == 0
    dref = (grad * gm if gamma else grad).to(torch.bfloat16)
    assert torch.equal(dos.cpu(), dref)
    close(db2, dref.float().sum(0), 1e-5, 1e-4)
    close(dg, (grad * y.float()).sum(0), 1e-5, 1e-4)
    d2 = torch.empty_like(dos)
    assert lib.cnx_scale_residual_bwd(gd.data_ptr(), 0, None, P(gmd), d2.data_ptr(), None, None, None, M_, C, S()) == 0
    assert torch.equal(d2, dos)
    assert lib.cnx_scale_residual_bwd(gd.data_ptr(), 0, None, P(gmd), d2.data_ptr(), dg.data_ptr(), None, ws.data_ptr(), M_, C, S()) == -1
    N = 4 * C
    dh = torch.randn(M_, N, generator=g_).to(torch.bfloat16)
    hp = (torch.randn(M_, N, generator=g_) * 2).to(torch.bfloat16)
    hpr = hp.float().requires_grad_()
    (ref,) = torch.autograd.grad(F.gelu(hpr), hpr, dh.float())
    dhp = torch.empty(M_, N, device="cuda", dtype=torch.bfloat16)
    db1 = torch.empty(N, device="cuda")
    assert lib.cnx_gelu_bwd_colsum(dh.cuda().data_ptr(), hp.cuda().data_ptr(), dhp.data_ptr(), db1.data_ptr(), ws.data_ptr(), M_,
                                   N, S()) == 0
    close(dhp, ref, 8e-3, 1e-4)
    close(db1, dhp.float().sum(0), 1e-5, 1e-4)


def test_derived_weight_copies_follow_a_fused_optimizer_step(R):
    """AdamW(fused=True) updates parameters in place WITHOUT bumping Tensor._version; the packed / bf16 weight copies of
    the fused kernels must still be rebuilt (global optimizer-step hook -> ops.invalidate_weight_cache)."""
    torch.manual_seed(0)
    blk = R.architecture.ConvNeXtBlock(96, ls_init_value=0.5).cuda()
    x = torch.randn(2, 96, 14, 14, device="cuda").contiguous(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(blk.parameters(), lr=0.05, fused=True)

    def both():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fused = blk(x)
            saved, R.ops.MODE = R.ops.MODE, "eager"
            try:
                eager = blk(x)
            finally:
                R.ops.MODE = saved
        return fused.float(), eager.float()

    f0, e0 = both()
    assert float((f0 - e0).norm() / (e0 - x).norm()) < 2e-2
    with torch.autocast("cuda", dtype=torch.bfloat16):
        blk(x).float().square().mean().backward()
    v = blk.mlp.fc1.weight._version
    opt.step()
    f1, e1 = both()
    assert float((e1 - e0).norm()) > 1e-3 * float(e0.norm())             # the step really moved the block
    assert float((f1 - e1).norm() / (e1 - x).norm()) < 2e-2, "fused path still uses the pre-step weights"


@pytest.mark.parametrize("P,N,H,W", [(48, 2, 32, 32), (64, 1, 16, 24), (96, 3, 8, 8), (48, 1, 224, 224)])
def test_stem_image_conv_forward_and_input_gradient(R, P, N, H, W):
    """cnx_stem_conv_fwd / cnx_stem_conv_dgrad (fp32 NCHW image <-> NHWC bf16 rows, stride 2) vs F.conv2d on the bf16-rounded
    operands (autocast numerics); the autograd wrapper's weight / bias gradients vs fp32 autograd."""
    g = torch.Generator().manual_seed(P + H)
    x = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(P, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(P, generator=g) * 0.1
    bf = lambda t: t.to(torch.bfloat16).float()
    xr, wr, br = bf(x).requires_grad_(), bf(w).requires_grad_(), b.clone().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride=2, padding=1)
    xd = x.cuda().requires_grad_()
    wd, bd = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert R.ops.stem_conv_supported(xd, wd, (2, 2), (1, 1))
        out = R.ops.stem_conv(xd, wd, bd)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    close(out, ref, 1e-2, 1e-2)                                   # bf16 output rounding
    cot = bf(torch.randn(ref.shape, generator=g))
    gx, gw, gb = torch.autograd.grad(ref, (xr, wr, br), cot)
    dx, dw, db = torch.autograd.grad(out, (xd, wd, bd), cot.cuda().to(torch.bfloat16))
    close(dx, gx, 1e-4, 1e-4 * float(gx.abs().max()))            # exact products of bf16 values, fp32 accumulation
    assert float((dw.float().cpu() - gw).norm() / gw.norm()) < 1e-2
    assert float((db.float().cpu() - gb).norm() / gb.norm()) < 1e-2
    with R.ops.input_grad_only():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out2 = R.ops.stem_conv(xd, wd, bd)
        (dx2,) = torch.autograd.grad(out2, xd, cot.cuda().to(torch.bfloat16))
    assert torch.equal(dx2, dx)


@pytest.mark.gpu
@pytest.mark.parametrize("P,N,H,W", [(48, 2, 32, 32), (64, 1, 16, 24), (96, 3, 8, 8), (48, 3, 224, 224), (64, 5, 64, 96), (48, 1, 2, 2),
                                     (96, 2, 32, 64)])
def test_stem_conv_filter_gradient_vs_fp32_reference(R, P, N, H, W):
    """cnx_stem_conv_wgrad (csrc/wgrad_kernels.hip: contraction over the positions, dy rows through LDS transpose reads) against the
    fp32 filter / bias gradient of F.conv2d on the operands the autocast convolution multiplies (x and dy rounded to bf16).  Widths
    whose output rows are multiples of 16 take the row-chunk form, the others the general one; sizes below one 16-position chunk
    and tails are covered.  Asymmetric random data: a transposed or permuted operand cannot pass."""
    lib = R._lib.load()
    g = torch.Generator().manual_seed(P + H)
    x = torch.rand(N, 3, H, W, generator=g)
    dy = torch.randn(N, P, H // 2, W // 2, generator=g) * torch.linspace(0.5, 2.0, P).view(1, P, 1, 1)
    xb, dyb = x.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()
    w = torch.zeros(P, 3, 3, 3, requires_grad=True)
    b = torch.zeros(P, requires_grad=True)
    gw, gb = torch.autograd.grad(F.conv2d(xb, w, b, stride=2, padding=1), (w, b), dyb)
    xd = x.cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    dw = torch.full((P, 3, 3, 3), float("nan"), device="cuda")
    db = torch.full((P,), float("nan"), device="cuda")
    ws = torch.empty(lib.cnx_stem_conv_wgrad_ws_floats(P), device="cuda")
    for dbp in (db, None):
        assert lib.cnx_stem_conv_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None if dbp is None else dbp.data_ptr(),
                                       ws.data_ptr(), N, H, W, P, S()) == 0
        scale = float(gw.abs().max()) + 1e-6
        close(dw, gw, 1e-4, 2e-5 * scale)                       # fp32 accumulation of exact bf16 products, another order
        close(db, gb, 1e-4, 2e-5 * float(gb.abs().max()) + 1e-6)
    dw2 = torch.empty_like(dw)
    assert lib.cnx_stem_conv_wgrad(xd.data_ptr(), dyd.data_ptr(), dw2.data_ptr(), None, ws.data_ptr(), N, H, W, P, S()) == 0
    assert torch.equal(dw, dw2)                                  # deterministic (fixed-order partial sums)
    assert lib.cnx_stem_conv_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), N, H + 1, W, P, S()) == -4
    assert lib.cnx_stem_conv_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), N, H, W, 40, S()) == -4


@pytest.mark.gpu
@pytest.mark.parametrize("P,N,H,W", [(48, 2, 32, 32), (64, 1, 16, 24), (96, 3, 8, 8), (48, 2, 224, 224)])
def test_stem_dgrad_sign_is_the_sign_of_the_stem_dgrad(R, P, N, H, W):
    """cnx_stem_conv_dgrad_sign stores sign() of exactly the fp32 value cnx_stem_conv_dgrad stores (zero rows included)."""
    lib = R._lib.load()
    g = torch.Generator().manual_seed(P + H)
    w = (torch.randn(P, 3, 3, 3, generator=g) * 0.3).cuda()
    wq = R.ops._pack_stem(w)
    dy = torch.randn(N, H // 2, W // 2, P, generator=g).to(torch.bfloat16).cuda()
    dy[0, 0] = 0                                                          # a band of exactly-zero gradients
    dx = torch.empty(N, 3, H, W, device="cuda")
    sg = torch.full((N, 3, H, W), 9, device="cuda", dtype=torch.int8)
    assert lib.cnx_stem_conv_dgrad(dy.data_ptr(), wq.data_ptr(), dx.data_ptr(), N, H, W, P, S()) == 0
    assert lib.cnx_stem_conv_dgrad_sign(dy.data_ptr(), wq.data_ptr(), sg.data_ptr(), N, H, W, P, S()) == 0
    assert torch.equal(sg, torch.sign(dx).to(torch.int8))
    assert int((sg == 0).sum()) > 0 and int((sg == 1).sum()) > 0 and int((sg == -1).sum()) > 0
    # the blocked order for the Linf update kernel (APGD_I8_BLK): the same signs, permuted inside 1024-element groups per sample
    sb = torch.full((N, 3, H, W), 9, device="cuda", dtype=torch.int8)
    rc = lib.cnx_stem_conv_dgrad_sign_blk(dy.data_ptr(), wq.data_ptr(), sb.data_ptr(), N, H, W, P, S())
    if (3 * H * W) % 1024 == 0:
        assert rc == 0
        sb.apgd_blocked = True
        assert torch.equal(R.ops.signs_to_linear(sb), sg)
        assert torch.equal(R.ops.signs_to_blocked(sg), sb)
    else:
        assert rc == -4


@pytest.mark.gpu
@pytest.mark.parametrize("soft", [False, True])
def test_attack_with_the_gradient_sign_sink_equals_the_attack_without(R, monkeypatch, soft):
    """apgd_train on the product ConvNeXt-T-CvSt with the int8 sign sink (default) and with fp32 gradients: every output
    identical bit for bit (Linf) - as long as the model's backward is run-to-run reproducible on this box, which is probed
    first (the library's convolution backward-data is not always: tools/determinism_check.py); otherwise >= 99 % agreement."""
    torch.manual_seed(4)
    model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("gamma"):
                p.fill_(0.5)
    model.eval()
    x = torch.rand(4, 3, 64, 64, device="cuda")
    y = torch.softmax(torch.randn(4, 1000, device="cuda"), 1) if soft else torch.tensor([1, 2, 3, 4], device="cuda")
    def input_grad():
        xr = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o = model(xr)
        with R.ops.input_grad_only():
            return torch.autograd.grad([o], [xr], grad_outputs=[torch.ones_like(o)])[0]

    probe = input_grad()
    reproducible = all(torch.equal(probe, input_grad()) for _ in range(8))

    def same(a, b, frac=0.99):
        return torch.equal(a, b) if reproducible else float((a == b).float().mean()) >= frac

    outs = {}
    for use in (True, False):
        monkeypatch.setattr(R.apgd, "USE_SIGN_SINK", use)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs[use] = R.apgd_train(model, x, y, norm="Linf", eps=4 / 255, n_iter=3, mixup=object() if soft else None)
    assert same(outs[True][0], outs[False][0]) and same(outs[True][3], outs[False][3])
    if reproducible:
        for a, b in zip(outs[True], outs[False]):
            assert torch.equal(a, b)
    # the sink really is used on this model (an int8 gradient reaches the update kernel) ...
    xi = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(xi)
    with R.ops.input_grad_only(), R.ops.grad_sign_sink(xi) as sk:
        (gz,) = torch.autograd.grad([out], [xi], grad_outputs=[torch.ones_like(out)])
    assert sk.signs is not None and sk.signs.dtype == torch.int8 and float(gz.abs().max()) == 0.0
    xj = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out2 = model(xj)
    with R.ops.input_grad_only():
        (g2,) = torch.autograd.grad([out2], [xj], grad_outputs=[torch.ones_like(out2)])
    assert same(sk.signs, torch.sign(g2).to(torch.int8), 0.999)
    # ... and a sink opened for a different tensor is left alone
    # (the forward runs where g2's did - outside input_grad_only: a forward that knows no parameter gradient will be asked for takes the
    #  attack's kernels, one that does not takes the training pass's, and the two round differently)
    with R.ops.grad_sign_sink(x.clone()) as other:
        xk = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out3 = model(xk)
        with R.ops.input_grad_only():
            (g3,) = torch.autograd.grad([out3], [xk], grad_outputs=[torch.ones_like(out3)])
    assert other.signs is None and same(torch.sign(g3), torch.sign(g2), 0.999)


@pytest.mark.gpu
@pytest.mark.parametrize("S,shape", [(1, (8,)), (3, (5, 8)), (64, (96, 40)), (17, (384, 96)), (512, (24, 16))])
def test_sum_parts_bf16_is_the_fp32_sum_of_the_partials(R, S, shape):
    """cnx_sum_parts_bf16 (split-K partial products of the weight-gradient GEMMs) vs torch's fp32-accumulated sum; the wrapper
    falls back to torch for layouts the kernel does not take."""
    g = torch.Generator().manual_seed(S)
    part = torch.randn(S, *shape, generator=g).to(torch.bfloat16).cuda()
    out = R.ops._sum_parts(part)
    assert out.dtype == torch.float32 and out.shape == part.shape[1:]
    ref = part.double().sum(0)
    close(out, ref, 1e-6, 1e-5)
    lib = R._lib.load()
    assert lib.cnx_sum_parts_bf16(None, out.data_ptr(), S, 8, torch.cuda.current_stream().cuda_stream) == -1
    assert lib.cnx_sum_parts_bf16(part.data_ptr(), out.data_ptr(), S, 12, torch.cuda.current_stream().cuda_stream) != 0
    odd = torch.randn(4, 3, 5).to(torch.bfloat16).cuda()                  # 15 elements per part: torch path
    close(R.ops._sum_parts(odd), odd.double().sum(0), 1e-6, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Co,H,W,xdt", [(2, 96, 192, 8, 12, torch.float32), (3, 192, 384, 6, 6, torch.float32),
                                            (1, 384, 768, 14, 14, torch.bfloat16), (2, 48, 64, 4, 2, torch.float32)])
def test_downsample_ln_patch_gemm_vs_fp32_reference(R, N, C, Co, H, W, xdt):
    """LayerNorm2d + Conv2d(k=2, s=2) as cnx_layernorm_fwd_patch2 + GEMM (+ cnx_layernorm_bwd_patch2) vs the fp32 eager pair:
    output, input gradient and all four parameter gradients; bf16 bar 1e-2 relative (Frobenius)."""
    g = torch.Generator().manual_seed(C + H)
    ln = torch.nn.LayerNorm(C, eps=1e-6)
    conv = torch.nn.Conv2d(C, Co, 2, 2)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(C, generator=g)); ln.bias.copy_(0.2 * torch.randn(C, generator=g))
    x = torch.randn(N, C, H, W, generator=g).to(xdt).float()
    gy = torch.randn(N, Co, H // 2, W // 2, generator=g)
    xr = x.clone().requires_grad_()
    ref = conv(ln(xr.permute(0, 2, 3, 1)).permute(0, 3, 1, 2))
    ref.backward(gy)
    refs = [ref.detach(), xr.grad, ln.weight.grad.clone(), ln.bias.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()]
    for p in list(ln.parameters()) + list(conv.parameters()):
        p.grad = None
    ln, conv = ln.cuda(), conv.cuda()
    xd = x.to(xdt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    if xdt == torch.float32:
        assert R.ops.downsample_supported(xd, ln.weight, conv) is False    # no autocast: fp32 activations stay in the library
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert R.ops.downsample_supported(xd, ln.weight, conv)
        out = R.ops.downsample_ln_conv(xd, ln.weight, ln.bias, 1e-6, conv.weight, conv.bias)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    out.backward(gy.cuda().to(out.dtype))
    got = [out.detach(), xd.grad, ln.weight.grad, ln.bias.grad, conv.weight.grad, conv.bias.grad]
    for a, b in zip(got, refs):
        err = float((a.float().cpu() - b).norm() / b.norm())
        assert err < 1e-2, err
    lib = R._lib.load()
    t = torch.zeros(64, device="cuda")
    assert lib.cnx_layernorm_fwd_patch2(t.data_ptr(), 0, t.data_ptr(), t.data_ptr(), 1e-6, t.data_ptr(), 1, None, None, 1, 3, 2, 96,
                                        torch.cuda.current_stream().cuda_stream) != 0          # odd H
    assert lib.cnx_layernorm_fwd_patch2(t.data_ptr(), 0, t.data_ptr(), t.data_ptr(), 1e-6, t.data_ptr(), 1, None, None, 1, 2, 2, 100,
                                        torch.cuda.current_stream().cuda_stream) != 0          # width without a wide kernel


@pytest.mark.gpu
@pytest.mark.parametrize("shape,gamma,xdt", [((3, 7, 96), True, torch.float32), ((2, 197, 768), False, torch.float32),
                                             ((5, 64), True, torch.bfloat16)])
def test_scale_residual_autograd_vs_eager(R, shape, gamma, xdt):
    """ops.scale_residual (cnx_scale_residual / cnx_scale_residual_bwd behind autograd): x + gamma*y and its three gradients
    vs the fp32 eager expression on the same bf16-rounded branch."""
    g = torch.Generator().manual_seed(shape[-1])
    x = torch.randn(*shape, generator=g).to(xdt)
    y = torch.randn(*shape, generator=g).to(torch.bfloat16)
    gm = torch.randn(shape[-1], generator=g) if gamma else None
    go = torch.randn(*shape, generator=g)
    xr, yr = x.float().clone().requires_grad_(), y.float().clone().requires_grad_()
    gr = gm.clone().requires_grad_() if gamma else None
    ref = xr + (yr * gr if gamma else yr)
    ref.backward(go)
    xd, yd = x.detach().cuda().requires_grad_(), y.detach().cuda().requires_grad_()
    gd = gm.cuda().requires_grad_() if gamma else None
    out = R.ops.scale_residual(xd, yd, gd)
    assert out.dtype == torch.float32
    out.backward(go.cuda())
    close(out, ref, 1e-6, 1e-6)
    close(xd.grad, xr.grad, 1e-2 if xdt == torch.bfloat16 else 1e-6, 1e-2 if xdt == torch.bfloat16 else 1e-6)
    close(yd.grad, yr.grad, 8e-3, 1e-3)                                    # branch gradient leaves as bf16
    if gamma:
        close(gd.grad, gr.grad, 2e-3, 2e-3 * float(gr.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("P,N,H,W", [(48, 2, 32, 32), (64, 1, 16, 24), (96, 3, 8, 8), (48, 1, 224, 224)])
def test_stem_conv_ln_gelu_fused_equals_the_two_kernel_composition(R, P, N, H, W):
    """cnx_stem_conv_ln_gelu_fwd (LayerNorm + GELU on the convolution tile in LDS) vs stem_conv followed by the LayerNorm+GELU
    kernel: same activation up to one bf16 ulp, same gradients w.r.t. image, filter, bias and LayerNorm parameters; a
    gradient-free forward (no convolution output written) gives the same activation."""
    g = torch.Generator().manual_seed(P * 3 + H)
    x = torch.rand(N, 3, H, W, generator=g).cuda()
    w = (torch.randn(P, 3, 3, 3, generator=g) * 0.3).cuda()
    b = (torch.randn(P, generator=g) * 0.1).cuda()
    lw = (1 + 0.2 * torch.randn(P, generator=g)).cuda()
    lb = (0.2 * torch.randn(P, generator=g)).cuda()
    cot = torch.randn(N, P, H // 2, W // 2, generator=g).to(torch.bfloat16).cuda()

    def run(fused):
        leaves = [t.clone().requires_grad_() for t in (x, w, b, lw, lb)]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                out = R.ops.stem_conv_ln_gelu(*leaves, 1e-6)
            else:
                out = R.ops.layer_norm_cf_gelu(R.ops.stem_conv(leaves[0], leaves[1], leaves[2]), leaves[3], leaves[4], 1e-6)
        grads = torch.autograd.grad(out, leaves, cot)
        return out.detach(), grads

    o_f, g_f = run(True)
    o_r, g_r = run(False)
    assert o_f.dtype == torch.bfloat16 and o_f.shape == o_r.shape
    close(o_f, o_r, 8e-3, 2e-3)
    for a, r in zip(g_f, g_r):
        assert float((a.float() - r.float()).norm() / r.float().norm()) < 5e-3
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        o_n = R.ops.stem_conv_ln_gelu(x, w, b, lw, lb, 1e-6)
    assert torch.equal(o_n, o_f)


@pytest.mark.gpu
@pytest.mark.parametrize("arch,nb,res", [("convnext_tiny", 2, 224), ("convnext_tiny", 3, 160), ("convnext_tiny", 1, 96),
                                         ("convnext_base", 2, 224), ("vit_s", 2, 224), ("deit_s", 2, 224), ("vit_b", 2, 224),
                                         # row counts that are multiples of 64 at every stage: the round-5 training pass (Hpre kernel
                                         # pair at C = 192 / 384 resp. 128 / 256, accumulator-order emit at C = 96, cnx_gemm_tn weight gradients)
                                         ("convnext_tiny", 64, 64), ("convnext_base", 64, 64), ("convnext_tiny", 16, 224)])
def test_train_step_gradients_hip_vs_library_composition(R, monkeypatch, arch, nb, res):
    """End to end at the benchmark's shapes (ConvNeXt-T-CvSt, 224x224, bf16 autocast, batch 2): logits, input gradient and
    EVERY parameter gradient of the hand-written path (rolling / tile depthwise kernels, fused LN+MLP blocks and their emit
    backward, library-GEMM blocks with the one-pass tails, patch-form downsample, split-K weight gradients, stem kernels)
    against the same model run as the plain library composition (ops.MODE = "eager") on the same weights and input."""
    torch.manual_seed(0)
    model = R.get_new_model(arch, pretrained=False, not_original=True).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():                                   # layer-scale at its init (1e-6) would hide the block branches
        for n_, p in model.named_parameters():
            if n_.endswith("gamma"):
                p.fill_(0.5)
    model.train()
    x = torch.rand(nb, 3, res, res, device="cuda")
    y = (torch.arange(nb, device="cuda") * 37 + 3) % 1000

    def run(mode):
        monkeypatch.setattr(R.ops, "MODE", mode)
        R.ops.invalidate_weight_cache()
        for p in model.parameters():
            p.grad = None
        xi = x.clone().requires_grad_()
        torch.clear_autocast_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(xi)
            loss = F.cross_entropy(logits.float(), y)
        loss.backward()
        return logits.detach().float(), xi.grad.clone(), {n_: p.grad.detach().float().clone() for n_, p in model.named_parameters()}

    lo_h, gx_h, gp_h = run("hip")
    lo_e, gx_e, gp_e = run("eager")
    assert float((lo_h - lo_e).norm() / lo_e.norm()) < 3e-2
    assert float((gx_h - gx_e).norm() / gx_e.norm()) < 6e-2
    worst = max((float((gp_h[k] - gp_e[k]).norm() / (gp_e[k].norm() + 1e-12)), k) for k in gp_e if float(gp_e[k].norm()) > 0)
    assert worst[0] < 8e-2, worst                           # two bf16 executions of a 60-kernel-deep chain
    assert set(gp_h) == set(gp_e) and all(torch.isfinite(v).all() for v in gp_h.values())


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,C", [(2, 14, 14, 64), (1, 56, 56, 32), (3, 7, 7, 96), (2, 28, 28, 64)])
@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_dwconv_add_operand_is_never_dropped_for_a_bf16_result(R, N, H, W, C, xdt):
    """The fused "+ add" operand of cnx_dwconv7x7_nhwc (the residual gradient riding into the input-gradient call) with a bf16
    result - the gradient of a block whose input is bf16, i.e. the first block of every stage - must equal the fp32 result
    rounded once.  (The packed-dot kernels fuse the add only into fp32 results; this combination used to lose the operand.)"""
    lib = R._lib.load()
    g = torch.Generator().manual_seed(N * H + C)
    d_u = torch.randn(N, H, W, C, generator=g).to(xdt).cuda()
    add = torch.randn(N, H, W, C, generator=g).cuda()
    w49 = (torch.randn(49, C, generator=g) * 0.1).cuda()
    o16 = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(N, H, W, C, device="cuda")
    code = R._lib.dtype_code(xdt)
    assert lib.cnx_dwconv7x7_nhwc(d_u.data_ptr(), code, w49.data_ptr(), None, add.data_ptr(), o16.data_ptr(), 1, N, H, W, C, 1, S()) == 0
    assert lib.cnx_dwconv7x7_nhwc(d_u.data_ptr(), code, w49.data_ptr(), None, add.data_ptr(), o32.data_ptr(), 0, N, H, W, C, 1, S()) == 0
    assert float((o32 - add).norm()) > 0.1 * float(add.norm())            # the stencil part is not negligible ...
    close(o16, o32, 8e-3, 8e-3)                                             # ... and the sum is only rounded


@pytest.mark.gpu
@pytest.mark.parametrize("norm,eps", [("Linf", 4 / 255), ("L2", 2.0)])
def test_apgd_on_the_hip_model_agrees_with_apgd_on_the_library_composition(R, monkeypatch, norm, eps):
    """The attack consumes only sign(gradient), so a wrongly scaled or partially missing input gradient does not show in the
    APGD state machine tests.  Here the same ConvNeXt-T-CvSt weights are attacked through the hand-written model path and through
    the plain library composition (bf16 autocast both): the adversarial images must agree except where the gradient is within
    bf16 noise of zero."""
    torch.manual_seed(1)
    model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("gamma"):
                p.fill_(0.5)
    model.eval()
    x = torch.rand(4, 3, 224, 224, device="cuda")
    y = torch.tensor([1, 2, 3, 4], device="cuda")
    outs = {}
    for mode in ("hip", "eager"):
        monkeypatch.setattr(R.ops, "MODE", mode)
        R.ops.invalidate_weight_cache()
        torch.clear_autocast_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs[mode] = R.apgd_train(model, x, y, norm=norm, eps=eps, n_iter=2)[0]
    if norm == "Linf":
        same = (outs["hip"] == outs["eager"]).float().mean().item()
        assert same > 0.9, same
        assert float((outs["hip"] - x).abs().max()) <= eps + 1e-6
    else:                                                   # L2: the normalised gradient direction must agree
        d_h, d_e = (outs["hip"] - x).flatten(1), (outs["eager"] - x).flatten(1)
        cos = F.cosine_similarity(d_h, d_e, dim=1)
        assert float(cos.min()) > 0.95, cos
        assert float(d_h.norm(dim=1).max()) <= eps * (1 + 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("loss,n_iter", [("ce", 10), ("dlr", 1)])
def test_eval_attack_on_the_hip_model_agrees_with_the_library_composition(R, monkeypatch, loss, n_iter):
    """aa_eval.apgd_attack (no random start) through the hand-written model path vs the library composition: the adversarial
    images agree except where the gradient is within bf16 noise of zero.  DLR divides by the top-1 - top-3 logit gap and picks
    classes by rank, both at the bf16 resolution of a random-init head: it is compared after one iteration and only on samples
    whose top-3 ranking is the same in the two executions."""
    torch.manual_seed(2)
    model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("gamma"):
                p.fill_(0.5)
    model.eval()
    x = torch.rand(8, 3, 224, 224, device="cuda")
    outs, top3 = {}, {}
    for mode in ("hip", "eager"):
        monkeypatch.setattr(R.ops, "MODE", mode)
        R.ops.invalidate_weight_cache()
        torch.clear_autocast_cache()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            top3[mode] = model(x).float().topk(3).indices
    y = top3["eager"][:, 0]                                     # attack the predicted class: every point starts robust
    for mode in ("hip", "eager"):
        monkeypatch.setattr(R.ops, "MODE", mode)
        R.ops.invalidate_weight_cache()
        torch.clear_autocast_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs[mode] = R.aa_eval.apgd_attack(model, x, y, "Linf", 4 / 255, n_iter, loss, None, False)
    keep = (top3["hip"] == top3["eager"]).all(1) if loss == "dlr" else torch.ones(8, dtype=torch.bool, device="cuda")
    assert int(keep.sum()) >= 2
    xa_h, xa_e = outs["hip"][3][keep], outs["eager"][3][keep]   # x_best
    same = (xa_h == xa_e).float().mean().item()
    assert same > 0.8, same                                     # sign flips of near-zero gradients compound over the iterations
    assert float((outs["hip"][3] - x).abs().max()) <= 4 / 255 + 1e-6


@pytest.mark.parametrize("N,C,H,W", [(4, 768, 7, 7), (3, 96, 5, 9), (2, 1536, 10, 10)])
def test_global_pool_on_channels_last_rows_equals_the_mean_over_the_map(R, N, C, H, W):
    """ops.global_pool (the head's pool, ``x.mean((-2, -1), keepdim=True)``): same value, same gradient - written once, in rows - for
    an NCHW-shaped view of channels-last rows; any other layout takes torch's mean."""
    g = torch.Generator(device="cuda").manual_seed(C + H)
    rows = torch.randn(N, H, W, C, device="cuda", generator=g)
    x = rows.permute(0, 3, 1, 2).requires_grad_()
    y = R.ops.global_pool(x)
    ref = x.mean((-2, -1), keepdim=True)
    assert y.shape == ref.shape and float((y - ref).abs().max()) <= 1e-6
    gy = torch.randn_like(ref)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gr,) = torch.autograd.grad(ref, x, gy)
    assert float((gx - gr).abs().max()) <= 1e-7 and gx.permute(0, 2, 3, 1).is_contiguous()
    xc = torch.randn(N, C, H, W, device="cuda", generator=g)                   # NCHW-contiguous: torch's path
    assert torch.equal(R.ops.global_pool(xc), xc.mean((-2, -1), keepdim=True))
