"""cnx_gemm_nt (csrc/gemm_kernels.hip) against fp32 torch references of the same arithmetic: the four epilogues, the three tile
configurations, ragged M / N, strided operands, and the operators that route through it (``ops.mlp_residual``, ``ops.linear_lib``,
the library-path ConvNeXt block, the downsample layer) against their hipBLASLt composition (``APGD_GEMM=lib`` path)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R
    assert torch.cuda.is_available()
    R._lib.load()
    return R


def bf(t):
    return t.to(torch.bfloat16)


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


SHAPES = [(256 * 100 + 37, 1536, 384), # 606 tiles of 256 x 256 (N a multiple of 256, >= 512 tiles), ragged last row tile
          (256 * 120, 1344, 128),      # 840 tiles of 256 x 192: the other 8-wavefront configuration
          (12544, 768, 3072),          # 128-row tiles, long K
          (1000, 192, 64), (777, 196, 128), (33, 4, 64), (4099, 388, 448)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_bias_and_gelu_epilogues(R, M, N, K):
    O = R.ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    b = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ w.float().t() + bf(b).float()
    out = O._gemm_nt(a, w, O.EPI_BIAS, bias=b)
    assert out.dtype == torch.bfloat16 and rel(out, ref) <= 3e-3
    assert rel(O._gemm_nt(a, w, O.EPI_BIAS), a.float() @ w.float().t()) <= 3e-3
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = O._gemm_nt(a, w, O.EPI_BIAS_GELU, bias=b, z_out=z)
    assert torch.equal(z, out)                                           # the pre-activation is the bias epilogue's result
    assert rel(h, F.gelu(z.float())) <= 3e-3
    assert float((h.float() - F.gelu(z.float())).abs().max()) <= 2e-2 * float(z.float().abs().max())


@pytest.mark.parametrize("M,N,K", SHAPES[:5])
@pytest.mark.parametrize("rdt,odt", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_gemm_scale_residual_epilogue(R, M, N, K, rdt, odt):
    O = R.ops
    g = torch.Generator(device="cuda").manual_seed(7)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    b = torch.randn(N, device="cuda", generator=g)
    gamma = torch.rand(N, device="cuda", generator=g) + 0.5
    r = torch.randn(M, N, device="cuda", generator=g).to(rdt)
    y = bf(a.float() @ w.float().t() + bf(b).float())
    ref = r.float() + gamma * y.float()
    y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = O._gemm_nt(a, w, O.EPI_SCALE_RES, bias=b, gamma=gamma, resid=r, out_dtype=odt, z_out=y2)
    assert out.dtype == odt and rel(out, ref) <= (3e-3 if odt == torch.float32 else 6e-3)
    assert rel(y2, y) <= 3e-3
    out2 = O._gemm_nt(a, w, O.EPI_SCALE_RES, bias=b, resid=r, out_dtype=odt)          # gamma = 1
    assert rel(out2, r.float() + y.float()) <= (3e-3 if odt == torch.float32 else 6e-3)


@pytest.mark.parametrize("M,N,K", SHAPES[:5])
def test_gemm_gelu_grad_epilogue(R, M, N, K):
    O = R.ops
    g = torch.Generator(device="cuda").manual_seed(11)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    z = bf(torch.randn(M, N, device="cuda", generator=g) * 2)
    zf = z.float().requires_grad_()
    (gp,) = torch.autograd.grad(F.gelu(zf).sum(), zf)
    ref = bf(a.float() @ w.float().t()).float() * gp
    out = O._gemm_nt(a, w, O.EPI_GELU_GRAD, z_in=z)
    assert rel(out, ref) <= 4e-3


def test_gemm_strided_operands_and_argument_errors(R):
    O, lib = R.ops, R._lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    big = bf(torch.randn(500, 256, device="cuda", generator=g))
    a = big[:, 64:192]                                                   # row stride 256, K = 128, 16-byte aligned
    w = bf(torch.randn(96, 128, device="cuda", generator=g))
    assert O._gemm_ok(a, w)
    assert rel(O._gemm_nt(a, w), a.float() @ w.float().t()) <= 3e-3
    S = torch.cuda.current_stream().cuda_stream
    out = torch.empty(500, 96, device="cuda", dtype=torch.bfloat16)
    args = lambda K, N=96, ep=0, dd=1: (a.data_ptr(), 256, w.data_ptr(), 128, out.data_ptr(), N, dd, 500, N, K, ep, None, None, None, N, 0,
                                        None, None, N, S)
    assert lib.cnx_gemm_nt(*args(128)) == 0
    assert lib.cnx_gemm_nt(*args(100)) == -4                             # K % 64
    assert lib.cnx_gemm_nt(*args(128, N=94)) == -4                       # N % 4
    assert lib.cnx_gemm_nt(*args(128, ep=3)) == -1                       # GELU' without the pre-activation
    assert lib.cnx_gemm_nt(*args(128, ep=0, dd=0)) == -3                 # fp32 result only for the residual epilogue
    assert lib.cnx_gemm_nt(*args(128, ep=9)) == -4
    assert lib.cnx_gemm_nt_supported(10, 96, 128) == 1 and lib.cnx_gemm_nt_supported(10, 96, 96) == 0


def _grads(fn, inputs):
    out = fn()
    gen = torch.Generator(device="cuda").manual_seed(5)
    cot = torch.randn(out.shape, device="cuda", generator=gen).to(out.dtype)
    return out.detach(), torch.autograd.grad(out, inputs, cot)


@pytest.mark.parametrize("C,M", [(768, 12544 // 4), (384, 5000)])
def test_mlp_residual_and_linear_on_the_gemm_kernels_match_the_library_composition(R, monkeypatch, C, M):
    O = R.ops
    g = torch.Generator(device="cuda").manual_seed(C)
    xs = torch.randn(M, C, device="cuda", generator=g).requires_grad_()
    h = bf(torch.randn(M, C, device="cuda", generator=g)).requires_grad_()
    w1 = (torch.randn(4 * C, C, device="cuda", generator=g) * C ** -0.5).requires_grad_()
    b1 = (0.1 * torch.randn(4 * C, device="cuda", generator=g)).requires_grad_()
    w2 = (torch.randn(C, 4 * C, device="cuda", generator=g) * (4 * C) ** -0.5).requires_grad_()
    b2 = (0.1 * torch.randn(C, device="cuda", generator=g)).requires_grad_()
    gm = (0.5 + 0.1 * torch.randn(C, device="cuda", generator=g)).requires_grad_()
    ins = [xs, h, w1, b1, w2, b2, gm]
    res = {}
    for mode in ("hip", "lib"):
        monkeypatch.setattr(O, "_GEMM_MODE", mode)
        O.invalidate_weight_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            res[mode] = _grads(lambda: O.mlp_residual(xs, h, w1, b1, w2, b2, gm), ins)
    assert rel(res["hip"][0], res["lib"][0]) <= 2e-3
    for a, b, name in zip(res["hip"][1], res["lib"][1], "xs h w1 b1 w2 b2 gamma".split()):
        assert rel(a, b) <= 1.5e-2, (name, rel(a, b))
    # qkv-style linear
    w = (torch.randn(3 * C, C, device="cuda", generator=g) * C ** -0.5).requires_grad_()
    b = (0.1 * torch.randn(3 * C, device="cuda", generator=g)).requires_grad_()
    for mode in ("hip", "lib"):
        monkeypatch.setattr(O, "_GEMM_MODE", mode)
        O.invalidate_weight_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            res[mode] = _grads(lambda: O.linear_lib(h, w, b), [h, w, b])
    assert rel(res["hip"][0], res["lib"][0]) <= 2e-3
    for a, b_, name in zip(res["hip"][1], res["lib"][1], "x w b".split()):
        assert rel(a, b_) <= 1.5e-2, (name, rel(a, b_))


@pytest.mark.parametrize("C,HW,input_grad_only", [(768, 7, False), (768, 7, True), (384, 14, False), (512, 14, True)])
def test_library_path_block_on_the_gemm_kernels_matches_hipblaslt(R, monkeypatch, C, HW, input_grad_only):
    """The ConvNeXt block at widths / passes without a fused LN+MLP kernel (models/convnext.py:37-50): cnx_gemm_nt with fused
    epilogues vs the round-2 composition (hipBLASLt GEMMs between one-pass kernels), forward and every gradient."""
    O, A = R.ops, R.architecture
    torch.manual_seed(C + HW)
    blk = A.ConvNeXtBlock(C, ls_init_value=0.5).cuda()
    x = torch.randn(8, C, HW, HW, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    monkeypatch.setattr(O, "_FUSED_WIDTHS", set())
    monkeypatch.setattr(O, "_HPRE_WIDTHS", set())
    ins = [x] if input_grad_only else [x] + list(blk.parameters())
    res = {}
    for mode in ("hip", "lib"):
        monkeypatch.setattr(O, "_GEMM_MODE", mode)
        O.invalidate_weight_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if input_grad_only:
                with O.input_grad_only():
                    res[mode] = _grads(lambda: blk(x), ins)
            else:
                res[mode] = _grads(lambda: blk(x), ins)
    assert rel(res["hip"][0], res["lib"][0]) <= 2e-3
    for i, (a, b) in enumerate(zip(res["hip"][1], res["lib"][1])):
        assert rel(a, b) <= 2e-2, (i, rel(a, b))


@pytest.mark.parametrize("M,N1,N2", [(64, 384, 192), (640, 192, 384), (1024, 256, 128), (320, 128, 256), (4096, 384, 96), (1984, 96, 384),
                                     (256, 256, 64), (192, 64, 256), (12544, 768, 192), (6272, 1536, 384), (3136, 768, 3072),
                                     (50176, 384, 96), (2048, 512, 128), (128, 2304, 768)])
def test_gemm_tn_vs_fp32_reference_through_the_c_abi(M, N1, N2):
    """cnx_gemm_tn: D = A^T B (contraction over the rows of two row-major bf16 operands; LDS transpose reads) against the fp32
    product of the same bf16 values.  All eight tile shapes, one and many splits of M, strided operands (views into wider
    tensors), determinism.  The operands are random with per-column scales: a transposed / permuted fragment cannot pass."""
    import revisiting_at_amd as R
    lib = R._lib.load()
    assert lib.cnx_gemm_tn_supported(M, N1, N2) == 1
    g = torch.Generator(device="cuda").manual_seed(M + N1)
    pa, pb = 8 * (M % 3), 16                                        # leading dimensions wider than the operands
    Aw = (torch.randn(M, N1 + pa, device="cuda", generator=g) * torch.linspace(0.5, 2.0, N1 + pa, device="cuda")).to(torch.bfloat16)
    Bw = (torch.randn(M, N2 + pb, device="cuda", generator=g) * torch.linspace(2.0, 0.5, N2 + pb, device="cuda")).to(torch.bfloat16)
    A, B = Aw[:, :N1], Bw[:, :N2]
    ref = A.float().t() @ B.float()
    D = torch.full((N1, N2), float("nan"), device="cuda")
    ws = torch.empty(max(4, lib.cnx_gemm_tn_ws_floats(M, N1, N2)), device="cuda")
    S = torch.cuda.current_stream().cuda_stream
    assert lib.cnx_gemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), D.data_ptr(), ws.data_ptr(), M, N1, N2, S) == 0
    err = float((D - ref).norm() / ref.norm())
    assert err < 2e-6, err                                          # exact bf16 products, fp32 sums in another order
    D2 = torch.empty_like(D)
    assert lib.cnx_gemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), D2.data_ptr(), ws.data_ptr(), M, N1, N2, S) == 0
    assert torch.equal(D, D2)
    assert lib.cnx_gemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), D.data_ptr(), ws.data_ptr(), M + 8, N1, N2, S) == -4
    assert lib.cnx_gemm_tn_supported(M, N1 + 32, N2) in (0, 1) and lib.cnx_gemm_tn_supported(M, 1000, 768) == 0


def _rows_to_acc(x):
    """[M, N] -> CNX_TN_ACC tiles ([M/32][N/32] x 2 KiB; element (m, n) of a tile at byte 64 m + 32 ((n/4) % 2) + 8 (n/8) + 2 (n % 4))."""
    M, N = x.shape
    t = x.reshape(M // 32, 32, N // 32, 4, 2, 4)                    # (row tile, m, column tile, q, half, e): n = 8 q + 4 half + e
    return t.permute(0, 2, 1, 4, 3, 5).contiguous().reshape(-1)


@pytest.mark.parametrize("M,C", [(64, 96), (8192, 96), (448, 192), (12544, 192), (6272, 384), (50176, 384), (1024, 128), (3136, 256),
                                 (640, 64), (1536, 768)])
def test_gemm_tn_pair_vs_fp32_reference_and_the_two_single_launches(M, C):
    """cnx_gemm_tn_pair (round 6): both weight gradients of a block - dW1 = dHpre^T a with d(b1) = sum dHpre, dW2 = dO^T H with
    d(b2) = sum dO - from ONE launch whose splits of M are shared by the tiles of both problems (dW2 accumulated transposed, written
    back by the fixed-order sum).  Against the fp32 products of the same bf16 values, against two cnx_gemm_tn_ex launches (same
    products, another split of M: fp32 rounding apart), bit-for-bit repeatable; all four tile shapes, one and many splits, a row
    operand that is a view into a wider tensor."""
    import revisiting_at_amd as R
    lib = R._lib.load()
    N1, N2 = 4 * C, C
    assert lib.cnx_gemm_tn_pair_supported(M, N1, N2) == 1
    g = torch.Generator(device="cuda").manual_seed(M + C)
    sc1, sc2 = torch.linspace(0.5, 2.0, N1, device="cuda"), torch.linspace(2.0, 0.5, N2 + 16, device="cuda")
    dhp = (torch.randn(M, N1, device="cuda", generator=g) * sc1 + 0.1).to(torch.bfloat16)
    h = (torch.randn(M, N1, device="cuda", generator=g) * sc1.flip(0) + 0.2).to(torch.bfloat16)
    a_w = (torch.randn(M, N2 + 16, device="cuda", generator=g) * sc2).to(torch.bfloat16)
    do_w = (torch.randn(M, N2 + 16, device="cuda", generator=g) * sc2.flip(0) - 0.1).to(torch.bfloat16)
    a, do = a_w[:, :N2], do_w[:, 8:8 + N2]
    dhp_t, h_t = _rows_to_acc(dhp), _rows_to_acc(h)
    S = torch.cuda.current_stream().cuda_stream
    nws = lib.cnx_gemm_tn_pair_ws_floats(M, N1, N2)
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device="cuda")

    def run():
        dw1 = torch.full((N1, N2), float("nan"), device="cuda"); db1 = torch.full((N1,), float("nan"), device="cuda")
        dw2 = torch.full((N2, N1), float("nan"), device="cuda"); db2 = torch.full((N2,), float("nan"), device="cuda")
        assert lib.cnx_gemm_tn_pair(dhp_t.data_ptr(), a.data_ptr(), a.stride(0), h_t.data_ptr(), do.data_ptr(), do.stride(0), dw1.data_ptr(),
                                    db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M, N1, N2, S) == 0
        return dw1, db1, dw2, db2
    dw1, db1, dw2, db2 = run()
    # the same contraction on the other stage loop (ring of 32-row stages / two 64-row buffers): the same MFMA sequence per
    # accumulator - bit for bit
    was = lib.cnx_runtime_switch(5, -1)
    try:
        for mode in (0, 2, 3):
            lib.cnx_runtime_switch(5, mode)
            for x, y in zip(run(), (dw1, db1, dw2, db2)):
                assert torch.equal(x, y), mode
    finally:
        lib.cnx_runtime_switch(5, was)
    rel = lambda t, r: float((t - r).norm() / r.norm())
    assert rel(dw1, dhp.float().t() @ a.float()) < 2e-6 and rel(dw2, do.float().t() @ h.float()) < 2e-6
    assert rel(db1, dhp.float().sum(0)) < 2e-6 and rel(db2, do.float().sum(0)) < 2e-6
    for x, y in zip(run(), (dw1, db1, dw2, db2)):
        assert torch.equal(x, y)
    # the two single launches
    ws1 = torch.empty(max(lib.cnx_gemm_tn_ws_floats(M, N1, N2), lib.cnx_gemm_tn_ws_floats(M, N2, N1)), device="cuda")
    e1 = torch.empty(N1, N2, device="cuda"); c1 = torch.empty(N1, device="cuda")
    e2 = torch.empty(N2, N1, device="cuda"); c2 = torch.empty(N2, device="cuda")
    doc = do.contiguous()
    assert lib.cnx_gemm_tn_ex(dhp_t.data_ptr(), 0, 1, a.data_ptr(), a.stride(0), 0, e1.data_ptr(), c1.data_ptr(), ws1.data_ptr(), M, N1, N2, S) == 0
    assert lib.cnx_gemm_tn_ex(doc.data_ptr(), N2, 0, h_t.data_ptr(), 0, 1, e2.data_ptr(), c2.data_ptr(), ws1.data_ptr(), M, N2, N1, S) == 0
    assert rel(dw1, e1) < 2e-6 and rel(dw2, e2) < 2e-6 and rel(db1, c1) < 2e-6 and rel(db2, c2) < 2e-6
    # argument checks
    assert lib.cnx_gemm_tn_pair(dhp_t.data_ptr(), a.data_ptr(), a.stride(0), h_t.data_ptr(), do.data_ptr(), do.stride(0), dw1.data_ptr(),
                                None, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M, N1, N2, S) == -1
    assert lib.cnx_gemm_tn_pair(dhp_t.data_ptr(), a.data_ptr(), N2 - 8, h_t.data_ptr(), do.data_ptr(), do.stride(0), dw1.data_ptr(),
                                db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M, N1, N2, S) == -4
    assert lib.cnx_gemm_tn_pair(dhp_t.data_ptr(), a.data_ptr(), a.stride(0), h_t.data_ptr(), do.data_ptr(), do.stride(0), dw1.data_ptr(),
                                db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), M + 8, N1, N2, S) == -4
    assert lib.cnx_gemm_tn_pair_supported(M, 96, 384) == 0 and lib.cnx_gemm_tn_pair_ws_floats(M, 96, 384) == 0   # (tiles with WI >= WJ only)
    assert lib.cnx_gemm_tn_pair_supported(M, 1000, 768) == 0 and lib.cnx_gemm_tn_pair_supported(M + 8, N1, N2) == 0
