"""GPU parity tests (run with ``-m gpu`` on the MI355X box): HIP kernels called through the
C ABI (include/apgd_hip.h) versus the pinned CPU oracle (oracle/apgd_oracle.py) and the
golden trajectories recorded from the reference (tests/golden/apgd_*.npz).

Bar: bit-exact for the fp32 attack state, class indices and flags; 1e-5 relative / 3e-7
absolute for the per-sample loss (torch's own fp32 log-softmax carries that error); 1e-5 for
the L2 path (reduction order)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import bits_equal, golden_cases, load_golden
from oracle import apgd_oracle as O

pytestmark = pytest.mark.gpu

LINF = [c for c in golden_cases() if c.startswith("linf")]
L2 = [c for c in golden_cases() if c.startswith("l2")]
TIE_FREE = ["linf_k0", "linf_k1", "linf_k2", "linf_k3", "linf_cl_k3", "linf_flat_k5", "linf_scripted_k4", "linf_soft_k2",
            "linf_dlr_k5"]


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R
    assert torch.cuda.is_available()
    R._lib.load()          # fails loudly if the HIP extension is missing
    return R


@pytest.fixture(scope="module")
def lib(R):
    return R._lib.load()


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def S():
    return torch.cuda.current_stream().cuda_stream


def sha(t):
    return hashlib.sha256(t.detach().float().contiguous().cpu().numpy().tobytes()).hexdigest()


# --------------------------------------------------------------------------- K0 prologue
@pytest.mark.parametrize("n", [0, 1, 3, 4, 1023, 4096 + 5, 6 * 3 * 224 * 224])
def test_init_clamp_and_copies(lib, n):
    rng = np.random.default_rng(n)
    x = (rng.random(n, dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    xd = dev(x) if n else torch.empty(0, device="cuda")
    a, b, c = (torch.full_like(xd, 7.0) for _ in range(3))
    assert lib.apgd_init_f32(xd.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), n, S()) == 0
    want = np.clip(x, 0, 1)
    for t in (a, b, c):
        assert bits_equal(t.cpu().numpy(), want)


def test_init_unaligned_and_null(lib):
    x = torch.rand(1001, device="cuda") * 2 - 0.5
    a = torch.zeros(1001, device="cuda")
    assert lib.apgd_init_f32(x[1:].data_ptr(), a[1:].data_ptr(), None, None, 1000, S()) == 0
    assert torch.equal(a[1:], x[1:].clamp(0, 1)) and a[0] == 0
    assert lib.apgd_init_f32(None, a.data_ptr(), None, None, 10, S()) == -1
    assert lib.apgd_init_f32(x.data_ptr(), a.data_ptr(), None, None, -1, S()) == -2


# --------------------------------------------------------------------------- K1 Linf step
def _step_inputs(B, E, seed, eps):
    rng = np.random.default_rng(seed)
    x = rng.random((B, E), dtype=np.float32)
    x[:, : min(E, 7)] = np.array([0, 1, eps, 1 - eps, 0.5, 2 * eps, 1e-8], np.float32)[: min(E, 7)]
    xa = np.clip(x + rng.uniform(-eps, eps, (B, E)).astype(np.float32), 0, 1).astype(np.float32)
    xo = np.clip(x + rng.uniform(-eps, eps, (B, E)).astype(np.float32), 0, 1).astype(np.float32)
    g = (rng.standard_normal((B, E)) * 1e-3).astype(np.float32)
    g[rng.random((B, E)) < 0.02] = 0.0
    g[rng.random((B, E)) < 0.01] = -0.0
    g[rng.random((B, E)) < 0.005] = np.nan
    g[rng.random((B, E)) < 0.005] = 1e-42
    step = (np.float32(2 * eps) / np.float32(2.0) ** rng.integers(0, 4, B)).astype(np.float32)
    return x, xa, xo, g, step


@pytest.mark.parametrize("B,E", [(1, 4), (3, 50), (5, 3 * 12 * 12), (2, 3 * 64 * 64), (7, 1021), (4, 3 * 224 * 224)])
@pytest.mark.parametrize("a", [1.0, 0.75])
def test_linf_step_bit_exact(lib, B, E, a):
    eps = 4 / 255
    x, xa, xo, g, step = _step_inputs(B, E, B * 1000 + E, eps)
    if a == 1.0:
        xo = xa.copy()
    want = O.linf_step(x, xa, xo, g, step, eps, a)
    xd, xad, xod, gd, sd = map(dev, (x, xa, xo, g, step))
    out = torch.empty_like(xd)
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), gd.data_ptr(), 0, sd.data_ptr(),
                                  out.data_ptr(), None, B, E, eps, a, S()) == 0
    assert bits_equal(out.cpu().numpy(), want)
    if a == 1.0:    # iteration-0 form: x_adv_old IS x_adv (same pointer) -> the 3-stream kernel, same bits
        o0 = torch.zeros_like(xd)
        assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xad.data_ptr(), gd.data_ptr(), 0, sd.data_ptr(),
                                      o0.data_ptr(), None, B, E, eps, a, S()) == 0
        assert bits_equal(o0.cpu().numpy(), want)
    # launch-shape variants give the same bits
    for bps, un, nt in [(1, 1, 0), (3, 2, 1), (8, 4, 0), (2, 4, 1)]:
        o2 = torch.zeros_like(xd)
        assert lib.apgd_linf_step_f32_ex(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), gd.data_ptr(), 0,
                                         sd.data_ptr(), o2.data_ptr(), None, B, E, eps, a, bps, un, nt, S()) == 0
        assert bits_equal(o2.cpu().numpy(), want), (bps, un, nt)
    # bf16 gradient (only the sign is used) and bf16 copy of the output
    gb = gd.to(torch.bfloat16)
    want_b = O.linf_step(x, xa, xo, gb.float().cpu().numpy(), step, eps, a)
    o3 = torch.zeros_like(xd)
    ob = torch.zeros(B, E, device="cuda", dtype=torch.bfloat16)
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), gb.data_ptr(), 1, sd.data_ptr(),
                                  o3.data_ptr(), ob.data_ptr(), B, E, eps, a, S()) == 0
    assert bits_equal(o3.cpu().numpy(), want_b)
    assert torch.equal(ob, o3.to(torch.bfloat16))


@pytest.mark.parametrize("B,E", [(3, 50), (2, 3 * 64 * 64), (7, 1021), (4, 3 * 224 * 224)])
@pytest.mark.parametrize("a", [1.0, 0.75])
def test_linf_step_with_int8_gradient_signs_is_bit_identical(lib, B, E, a):
    """grad_dtype = APGD_I8: sign(grad) as int8 {-1, 0, +1} (sign(+-0) = sign(NaN) = 0) gives the bits of the fp32 call -
    the update reads nothing else of the gradient (autopgd_train_clean.py:221) - in every launch form."""
    eps = 4 / 255
    x, xa, xo, g, step = _step_inputs(B, E, B * 77 + E, eps)
    if a == 1.0:
        xo = xa.copy()
    want = O.linf_step(x, xa, xo, g, step, eps, a)
    sg = np.zeros(g.shape, np.int8)
    sg[g > 0] = 1
    sg[g < 0] = -1
    xd, xad, xod, sd = map(dev, (x, xa, xo, step))
    sgd = torch.from_numpy(sg).cuda()
    out = torch.zeros_like(xd)
    xo_ptr = xad.data_ptr() if a == 1.0 else xod.data_ptr()                    # a == 1: the 3-stream first-iteration form
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xo_ptr, sgd.data_ptr(), 3, sd.data_ptr(),
                                  out.data_ptr(), None, B, E, eps, a, S()) == 0
    assert bits_equal(out.cpu().numpy(), want)
    for bps, un, nt in [(1, 1, 0), (3, 2, 1), (8, 4, 0)]:
        o2 = torch.zeros_like(xd)
        assert lib.apgd_linf_step_f32_ex(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), sgd.data_ptr(), 3,
                                         sd.data_ptr(), o2.data_ptr(), None, B, E, eps, a, bps, un, nt, S()) == 0
        assert bits_equal(o2.cpu().numpy(), want), (bps, un, nt)
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), sgd.data_ptr(), 7, sd.data_ptr(),
                                  out.data_ptr(), None, B, E, eps, a, S()) == -3     # unknown dtype code


def test_track_rows_moves_int8_sign_rows(lib):
    B, E = 6, 3 * 16 * 16
    g = torch.Generator(device="cuda").manual_seed(0)
    flags = torch.tensor([0, 1, 2, 4, 5, 7], device="cuda", dtype=torch.uint8)
    xa = torch.rand(B, E, device="cuda", generator=g)
    xb, xba = torch.rand(B, E, device="cuda", generator=g), torch.rand(B, E, device="cuda", generator=g)
    gr = torch.randint(-1, 2, (B, E), device="cuda", generator=g, dtype=torch.int8)
    gb = torch.randint(-1, 2, (B, E), device="cuda", generator=g, dtype=torch.int8)
    xa0, xb0, gr0, gb0 = xa.clone(), xb.clone(), gr.clone(), gb.clone()
    assert lib.apgd_track_rows(flags.data_ptr(), xa.data_ptr(), gr.data_ptr(), xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), 1, B, E,
                               0, S()) == 0
    for b, f in enumerate(flags.tolist()):
        nb, rs = bool(f & 1), bool(f & 4) and not bool(f & 1)
        assert torch.equal(gb[b], gr0[b] if nb else gb0[b])
        assert torch.equal(gr[b], gb0[b] if rs else gr0[b])
        assert torch.equal(xb[b], xa0[b] if nb else xb0[b])
        assert torch.equal(xa[b], xb0[b] if rs else xa0[b])


@pytest.mark.parametrize("B,E", [(1, 1024), (3, 3 * 32 * 32), (5, 3 * 64 * 64), (2, 7 * 1024)])
@pytest.mark.parametrize("a", [1.0, 0.75])
def test_linf_step_with_blocked_int8_signs_is_bit_identical(lib, B, E, a):
    """grad_dtype = APGD_I8_BLK (include/apgd_hip.h): the signs permuted inside 1024-element groups give the bits of the element-order
    int8 call and of the oracle, in the general and the first-iteration form; sizes that are not whole groups are refused."""
    import revisiting_at_amd as R
    eps = 4 / 255
    x, xa, xo, g, step = _step_inputs(B, E, B * 31 + E, eps)
    if a == 1.0:
        xo = xa.copy()
    want = O.linf_step(x, xa, xo, g, step, eps, a)
    sg = np.zeros(g.shape, np.int8)
    sg[g > 0] = 1
    sg[g < 0] = -1
    xd, xad, xod, sd = map(dev, (x, xa, xo, step))
    sb = R.ops.signs_to_blocked(torch.from_numpy(sg).cuda())
    assert not torch.equal(sb, torch.from_numpy(sg).cuda()) and torch.equal(R.ops.signs_to_linear(sb).cpu(), torch.from_numpy(sg))
    out = torch.zeros_like(xd)
    xo_ptr = xad.data_ptr() if a == 1.0 else xod.data_ptr()
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xo_ptr, sb.data_ptr(), 4, sd.data_ptr(), out.data_ptr(), None, B, E,
                                  eps, a, S()) == 0
    assert bits_equal(out.cpu().numpy(), want)
    bad = torch.zeros(B, E + 4, device="cuda")
    assert lib.apgd_linf_step_f32(bad.data_ptr(), bad.data_ptr(), bad.data_ptr(), sb.data_ptr(), 4, sd.data_ptr(), out.data_ptr(), None,
                                  B, E + 4, eps, a, S()) == -4


def test_track_rows_moves_int8_sign_rows_of_odd_length(lib):
    """grad_elt = 1 with an odd row length (E = 3*5*5): every second row starts on an odd address - the copy must be byte
    exact, last byte of each row included (round-2 advice: the 2-byte fallback dropped it)."""
    B, E = 5, 3 * 5 * 5
    g = torch.Generator(device="cuda").manual_seed(4)
    flags = torch.tensor([1, 4, 5, 0, 1], device="cuda", dtype=torch.uint8)
    xa, xb, xba = (torch.rand(B, E, device="cuda", generator=g) for _ in range(3))
    gr = torch.randint(-1, 2, (B, E), device="cuda", generator=g, dtype=torch.int8)
    gb = torch.randint(-1, 2, (B, E), device="cuda", generator=g, dtype=torch.int8)
    gr0, gb0 = gr.clone(), gb.clone()
    assert lib.apgd_track_rows(flags.data_ptr(), xa.data_ptr(), gr.data_ptr(), xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), 1, B, E,
                               0, S()) == 0
    for b, f in enumerate(flags.tolist()):
        nb, rs = bool(f & 1), bool(f & 4) and not bool(f & 1)
        assert torch.equal(gb[b], gr0[b] if nb else gb0[b]), b
        assert torch.equal(gr[b], gb0[b] if rs else gr0[b]), b


@pytest.mark.parametrize("B,E", [(8, 3 * 16 * 16), (8, 3 * 5 * 5), (16, 1021), (9, 3 * 64 * 64), (256, 3 * 32 * 32)])
@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16, torch.int8])
def test_linf_step_track_equals_track_rows_then_step(lib, B, E, gdt):
    """apgd_linf_step_track_f32 (round 5: the Linf step of iteration i + 1 also performs the row moves of iteration i) against the
    two-pass sequence it replaces - apgd_track_rows, then apgd_linf_step_f32 - for every flag byte, bit for bit on EVERY buffer:
    the new iterate, x_best, grad_best, x_best_adv and the restored x_adv; the gradient buffer is the one deliberate difference
    (a restored row is not written back: nothing reads it before the next backward replaces it)."""
    eps, a = 4 / 255, 0.75
    x, xa, xo, g, step = _step_inputs(B, E, B * 13 + E, eps)
    gen = torch.Generator(device="cuda").manual_seed(E)
    flags = (torch.arange(B, device="cuda") % 8).to(torch.uint8)
    xd, xad, xod, sd = map(dev, (x, xa, xo, step))
    if gdt == torch.int8:
        gr = torch.randint(-1, 2, (B, E), device="cuda", generator=gen, dtype=torch.int8)
        gb = torch.randint(-1, 2, (B, E), device="cuda", generator=gen, dtype=torch.int8)
    else:
        gr = dev(g).to(gdt)
        gb = (torch.randn(B, E, device="cuda", generator=gen) * 1e-3).to(gdt)
    code = {torch.float32: 0, torch.bfloat16: 1, torch.int8: 3}[gdt]
    xb, xba = torch.rand(B, E, device="cuda", generator=gen), torch.rand(B, E, device="cuda", generator=gen)
    # reference: the two passes of rounds 1 - 4
    xa1, gr1, xb1, gb1, xba1 = xad.clone(), gr.clone(), xb.clone(), gb.clone(), xba.clone()
    out1 = torch.zeros_like(xd)
    assert lib.apgd_track_rows(flags.data_ptr(), xa1.data_ptr(), gr1.data_ptr(), xb1.data_ptr(), gb1.data_ptr(), xba1.data_ptr(),
                               gr.element_size(), B, E, 0, S()) == 0
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xa1.data_ptr(), xod.data_ptr(), gr1.data_ptr(), code, sd.data_ptr(),
                                  out1.data_ptr(), None, B, E, eps, a, S()) == 0
    xa2, gr2, xb2, gb2, xba2 = xad.clone(), gr.clone(), xb.clone(), gb.clone(), xba.clone()
    out2 = torch.zeros_like(xd)
    assert lib.apgd_linf_step_track_f32(xd.data_ptr(), xa2.data_ptr(), xod.data_ptr(), gr2.data_ptr(), code, sd.data_ptr(),
                                        out2.data_ptr(), flags.data_ptr(), xb2.data_ptr(), gb2.data_ptr(), xba2.data_ptr(),
                                        B, E, eps, a, S()) == 0
    for name, u, v in (("out", out1, out2), ("x_adv", xa1, xa2), ("x_best", xb1, xb2), ("x_best_adv", xba1, xba2)):
        assert bits_equal(u.cpu().numpy(), v.cpu().numpy()), name
    assert torch.equal(gb1.view(torch.uint8), gb2.view(torch.uint8))
    assert torch.equal(gr2.view(torch.uint8), gr.view(torch.uint8))          # the fused pass never writes the gradient


@pytest.mark.parametrize("B,E", [(4, 3 * 16 * 16), (3, 75), (2, 3 * 224 * 224)])
@pytest.mark.parametrize("gdt", [torch.float32, torch.int8])
def test_linf_step_track_first_iteration_makes_the_prologue_clones(lib, B, E, gdt):
    """flags == NULL: iteration 0 (x_adv_old is x_adv, a == 1) - the new iterate equals apgd_linf_step_f32's, and x_best =
    x_best_adv = x_adv (:142-143), grad_best = grad (:189) are written by the same launch."""
    eps = 4 / 255
    x, xa, _, g, step = _step_inputs(B, E, B * 5 + E, eps)
    xd, xad, sd = map(dev, (x, xa, step))
    gr = torch.sign(dev(g)).to(torch.int8) if gdt == torch.int8 else dev(g)
    code = 3 if gdt == torch.int8 else 0
    want = torch.zeros_like(xd)
    assert lib.apgd_linf_step_f32(xd.data_ptr(), xad.data_ptr(), xad.data_ptr(), gr.data_ptr(), code, sd.data_ptr(), want.data_ptr(),
                                  None, B, E, eps, 1.0, S()) == 0
    out, xb, xba, gb = torch.zeros_like(xd), torch.full_like(xd, -1), torch.full_like(xd, -1), torch.zeros_like(gr)
    xa0 = xad.clone()
    assert lib.apgd_linf_step_track_f32(xd.data_ptr(), xad.data_ptr(), xad.data_ptr(), gr.data_ptr(), code, sd.data_ptr(),
                                        out.data_ptr(), None, xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), B, E, eps, 1.0, S()) == 0
    assert bits_equal(out.cpu().numpy(), want.cpu().numpy())
    assert torch.equal(xb, xa0) and torch.equal(xba, xa0) and torch.equal(xad, xa0)
    assert torch.equal(gb.view(torch.uint8), gr.view(torch.uint8))
    # argument checks: the first form needs x_adv_old == x_adv and a == 1; aliasing outputs are refused
    assert lib.apgd_linf_step_track_f32(xd.data_ptr(), xad.data_ptr(), xd.data_ptr(), gr.data_ptr(), code, sd.data_ptr(),
                                        out.data_ptr(), None, xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), B, E, eps, 1.0, S()) == -4
    assert lib.apgd_linf_step_track_f32(xd.data_ptr(), xad.data_ptr(), xad.data_ptr(), gr.data_ptr(), code, sd.data_ptr(),
                                        out.data_ptr(), None, xb.data_ptr(), gb.data_ptr(), xb.data_ptr(), B, E, eps, 1.0, S()) == -4
    assert lib.apgd_linf_step_track_f32(xd.data_ptr(), xad.data_ptr(), xad.data_ptr(), gr.data_ptr(), 4, sd.data_ptr(),
                                        out.data_ptr(), None, xb.data_ptr(), gb.data_ptr(), xba.data_ptr(), B, E, eps, 1.0, S()) == -3


def test_linf_step_unaligned_rows(lib):
    eps, B, E = 8 / 255, 3, 64
    x, xa, xo, g, step = _step_inputs(B, E + 1, 5, eps)
    want = O.linf_step(x[:, 1:], xa[:, 1:], xo[:, 1:], g[:, 1:], step, eps, 0.75)
    flat = [dev(np.ascontiguousarray(t[:, 1:]).reshape(-1)) for t in (x, xa, xo, g)]
    pad = [torch.cat([torch.zeros(1, device="cuda"), t]) for t in flat]       # data starts 4 bytes off alignment
    out = torch.zeros(B * E + 1, device="cuda")
    sd = dev(step)
    assert lib.apgd_linf_step_f32(pad[0][1:].data_ptr(), pad[1][1:].data_ptr(), pad[2][1:].data_ptr(),
                                  pad[3][1:].data_ptr(), 0, sd.data_ptr(), out[1:].data_ptr(), None, B, E, eps, 0.75,
                                  S()) == 0
    assert bits_equal(out[1:].cpu().numpy().reshape(B, E), want)


def test_linf_step_argument_errors(lib):
    t = torch.zeros(16, device="cuda")
    p = t.data_ptr()
    assert lib.apgd_linf_step_f32(p, p, p, p, 0, p, None, None, 1, 16, 0.1, 1.0, S()) == -1
    assert lib.apgd_linf_step_f32(p, p, p, p, 0, p, p, None, 1, 16, 0.1, 1.0, S()) == -4      # out aliases an input
    o = torch.zeros(16, device="cuda")
    assert lib.apgd_linf_step_f32(p, p, p, p, 7, p, o.data_ptr(), None, 1, 16, 0.1, 1.0, S()) == -3
    assert lib.apgd_linf_step_f32(p, p, p, p, 0, p, o.data_ptr(), None, -1, 16, 0.1, 1.0, S()) == -2
    assert lib.apgd_linf_step_f32(p, p, p, p, 0, p, o.data_ptr(), None, 0, 16, 0.1, 1.0, S()) == 0   # empty batch


def test_linf_step_full_size_properties(lib):
    """BASELINE config #2 size (B=256, 3x224x224): bit-exact vs the oracle on a slice, and the
    eps-ball / box invariants (utils_eval.py:67-81) on everything via the device checker."""
    eps, B, E = 4 / 255, 256, 3 * 224 * 224
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(B, E, device="cuda", generator=gen)
    xa = (x + (torch.rand(B, E, device="cuda", generator=gen) * 2 - 1) * eps).clamp(0, 1)
    xo = (x + (torch.rand(B, E, device="cuda", generator=gen) * 2 - 1) * eps).clamp(0, 1)
    g = torch.randn(B, E, device="cuda", generator=gen) * 1e-3
    step = torch.full((B,), 2 * eps, device="cuda")
    step[::3] /= 2
    step[::5] /= 4
    out = torch.empty_like(x)
    assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), xo.data_ptr(), g.data_ptr(), 0, step.data_ptr(),
                                  out.data_ptr(), None, B, E, eps, 0.75, S()) == 0
    sl = slice(0, 256, 37)
    want = O.linf_step(*(t[sl].cpu().numpy() for t in (x, xa, xo, g, step)), eps, 0.75)
    assert bits_equal(out[sl].cpu().numpy(), want)
    chk = torch.empty(B, 3, device="cuda")
    assert lib.apgd_check_imgs_f32(out.data_ptr(), x.data_ptr(), chk.data_ptr(), B, E, S()) == 0
    chk = chk.cpu().numpy()
    assert chk[:, 0].max() <= np.float32(eps) * (1 + 2e-7) + 6e-8 and chk[:, 1].min() >= 0 and chk[:, 2].max() <= 1
    # idempotence of the projection: stepping with a zero gradient from a point already inside
    # the ball with x_adv_old == x_adv leaves it unchanged
    z = torch.zeros_like(g)
    o2 = torch.empty_like(x)
    assert lib.apgd_linf_step_f32(x.data_ptr(), out.data_ptr(), out.data_ptr(), z.data_ptr(), 0, step.data_ptr(),
                                  o2.data_ptr(), None, B, E, eps, 0.75, S()) == 0
    assert torch.equal(o2, out)


def test_linf_step_full_size_with_int8_gradient_signs(lib):
    """The PRODUCT form of K1 at BASELINE config #2's size (B = 256, 3x224x224): gradient signs as int8, both the general
    (a = 0.75) and the first-iteration (x_adv_old is x_adv) launch.  Bit-exact vs the oracle on a strided sample of rows AND vs
    the fp32-gradient launch on every element; eps-ball / box on everything."""
    eps, B, E = 4 / 255, 256, 3 * 224 * 224
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, E, device="cuda", generator=gen)
    xa = (x + (torch.rand(B, E, device="cuda", generator=gen) * 2 - 1) * eps).clamp(0, 1)
    xo = (x + (torch.rand(B, E, device="cuda", generator=gen) * 2 - 1) * eps).clamp(0, 1)
    g = torch.randn(B, E, device="cuda", generator=gen) * 1e-3
    g[:, ::7] = 0.0                                          # exact zeros: sign 0
    g[:, 5::11] = -0.0
    g[3, :100] = float("nan")                                # sign(NaN) = 0 in the reference's torch.sign
    sg = torch.sign(torch.nan_to_num(g, nan=0.0)).to(torch.int8)
    step = torch.full((B,), 2 * eps, device="cuda")
    step[::3] /= 2
    step[::5] /= 4
    for a, old in ((0.75, xo), (1.0, xa)):
        o8, o32 = torch.empty_like(x), torch.empty_like(x)
        assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), old.data_ptr(), sg.data_ptr(), 3, step.data_ptr(),
                                      o8.data_ptr(), None, B, E, eps, a, S()) == 0
        assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), old.data_ptr(), g.data_ptr(), 0, step.data_ptr(),
                                      o32.data_ptr(), None, B, E, eps, a, S()) == 0
        assert torch.equal(o8.view(torch.int32), o32.view(torch.int32)), a          # every element, bit for bit
        # ... and in the BLOCKED sign order the product's stem kernel writes (APGD_I8_BLK: 16 signs per lane in one load)
        import revisiting_at_amd as R
        sb = R.ops.signs_to_blocked(sg)
        assert torch.equal(R.ops.signs_to_linear(sb), sg)
        ob = torch.empty_like(x)
        assert lib.apgd_linf_step_f32(x.data_ptr(), xa.data_ptr(), old.data_ptr(), sb.data_ptr(), 4, step.data_ptr(),
                                      ob.data_ptr(), None, B, E, eps, a, S()) == 0
        assert torch.equal(ob.view(torch.int32), o32.view(torch.int32)), ("blocked", a)
        sl = slice(3, 256, 41)
        want = O.linf_step(*(t[sl].cpu().numpy() for t in (x, xa, old, g, step)), eps, a)
        assert bits_equal(o8[sl].cpu().numpy(), want), a
        chk = torch.empty(B, 3, device="cuda")
        assert lib.apgd_check_imgs_f32(o8.data_ptr(), x.data_ptr(), chk.data_ptr(), B, E, S()) == 0
        chk = chk.cpu().numpy()
        assert chk[:, 0].max() <= np.float32(eps) * (1 + 2e-7) + 6e-8 and chk[:, 1].min() >= 0 and chk[:, 2].max() <= 1


# --------------------------------------------------------------------------- K2 loss / pred / dlogits
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C", [(1, 2), (5, 10), (33, 1000), (256, 1000), (3, 63), (4, 65)])
@pytest.mark.parametrize("soft", [False, True])
def test_loss_pred_vs_oracle(R, lib, dtype, B, C, soft):
    rng = np.random.default_rng(B * 7 + C)
    z = (rng.standard_normal((B, C)) * 3).astype(np.float32)
    z[0, :] = 0.5                                   # all-equal row: argmax must be index 0
    if B > 2:
        z[2, C // 2] = z[2].max() + 1
        z[2, C - 1] = z[2, C // 2]                  # tie between two maxima: first index wins
    zt = dev(z).to(dtype)
    zf = zt.float().cpu().numpy()
    if soft:
        y = np.exp(rng.standard_normal((B, C))).astype(np.float32)
        y /= y.sum(1, keepdims=True)
        yd, yh, ys = dev(y), None, dev(y)
    else:
        y = rng.integers(0, C, B)
        y[0] = 0
        if B > 2:
            y[2] = C // 2
        yd, yh, ys = dev(y), dev(y), None
    loss = torch.empty(B, device="cuda")
    pred = torch.empty(B, device="cuda", dtype=torch.uint8)
    dl = torch.empty_like(zt)
    assert lib.apgd_loss_pred(zt.data_ptr(), R._lib.dtype_code(dtype), C, None if yh is None else yh.data_ptr(),
                              None if ys is None else ys.data_ptr(), 0, loss.data_ptr(), pred.data_ptr(),
                              dl.data_ptr(), B, C, S()) == 0
    np.testing.assert_allclose(loss.cpu().numpy(), O.ce_loss(zf, y), rtol=1e-5, atol=3e-7)
    assert np.array_equal(pred.cpu().numpy().astype(bool), O.predict(zf, y))
    tol = {torch.float32: 2e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    np.testing.assert_allclose(dl.float().cpu().numpy(), O.ce_dlogits(zf, y), atol=tol, rtol=tol)
    # against torch's own GPU cross-entropy + autograd (what the reference would run here)
    zt2 = zt.clone().requires_grad_()
    li = torch.nn.functional.cross_entropy(zt2.float(), yd, reduction="none")
    (gt,) = torch.autograd.grad(li.sum(), zt2)
    np.testing.assert_allclose(loss.cpu().numpy(), li.detach().cpu().numpy(), rtol=1e-5, atol=3e-7)
    np.testing.assert_allclose(dl.float().cpu().numpy(), gt.float().cpu().numpy(), atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C", [(1, 3), (7, 10), (64, 1000), (5, 65)])
def test_dlr_loss_pred_and_gradient(R, lib, dtype, B, C):
    """loss_kind 1 = dlr_loss (autopgd_train_clean.py:99-104): loss bit-exact vs the oracle (it is +,-,*,/ only),
    gradient vs torch autograd of the same formula on the CPU."""
    rng = np.random.default_rng(B + C)
    z = (rng.standard_normal((B, C)) * 3).astype(np.float32)
    y = rng.integers(0, C, B)
    y[0] = int(np.argmax(z[0]))                     # one correctly classified sample (ind = 1 branch)
    if B > 4:
        z[4, 1] = z[4, 2] = z[4].max() + 1          # tied maxima: stable-sort order decides `ind`
        y[4] = 2
    zt = dev(z).to(dtype)
    zf = zt.float().cpu()
    loss = torch.empty(B, device="cuda")
    pred = torch.empty(B, device="cuda", dtype=torch.uint8)
    dl = torch.empty_like(zt)
    yd = dev(y)
    assert lib.apgd_loss_pred(zt.data_ptr(), R._lib.dtype_code(dtype), C, yd.data_ptr(), None, 1, loss.data_ptr(),
                              pred.data_ptr(), dl.data_ptr(), B, C, S()) == 0
    assert bits_equal(loss.cpu().numpy(), O.dlr_loss(zf.numpy(), y))
    assert np.array_equal(pred.cpu().numpy().astype(bool), O.predict(zf.numpy(), y))
    zr = zf.clone().requires_grad_()
    zs, idx = zr.sort(dim=1)
    ind = (idx[:, -1] == torch.as_tensor(y)).float()
    u = torch.arange(B)
    ref = -(zr[u, torch.as_tensor(y)] - zs[:, -2] * ind - zs[:, -1] * (1. - ind)) / (zs[:, -1] - zs[:, -3] + 1e-12)
    (gr,) = torch.autograd.grad(ref.sum(), zr)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    # bf16 rounding creates accidental ties among the 4 largest logits; which of two equal entries receives the
    # gradient is then the sort's tie order (undefined for the reference's device sort) — compare the other rows,
    # plus row 4 whose tie is constructed and follows the stable order documented in the kernel
    top4 = zs.detach()[:, -4:] if C >= 4 else zs.detach()
    clean = ((top4[:, 1:] - top4[:, :-1]) != 0).all(dim=1).numpy()
    if B > 4:
        clean[4] = bool((top4[4, 1:-1] - top4[4, :-2] != 0).all())
    assert clean.sum() >= B // 2
    got = dl.float().cpu().numpy()
    gmax = float(gr[torch.as_tensor(clean)].abs().max())
    np.testing.assert_allclose(got[clean], gr.numpy()[clean], rtol=tol, atol=tol * gmax)
    assert lib.apgd_loss_pred(zt.data_ptr(), R._lib.dtype_code(dtype), C, None, dl.data_ptr(), 1, loss.data_ptr(),
                              pred.data_ptr(), None, B, C, S()) == -4          # dlr needs hard labels


def test_loss_pred_strided_rows_and_errors(R, lib):
    B, C, ld = 6, 10, 16
    buf = torch.randn(B, ld, device="cuda")
    y = torch.randint(0, C, (B,), device="cuda")
    loss = torch.empty(B, device="cuda")
    pred = torch.empty(B, device="cuda", dtype=torch.uint8)
    assert lib.apgd_loss_pred(buf.data_ptr(), 0, ld, y.data_ptr(), None, 0, loss.data_ptr(), pred.data_ptr(), None, B,
                              C, S()) == 0
    want = torch.nn.functional.cross_entropy(buf[:, :C], y, reduction="none")
    np.testing.assert_allclose(loss.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=3e-7)
    assert torch.equal(pred.bool(), buf[:, :C].argmax(1) == y)
    assert lib.apgd_loss_pred(buf.data_ptr(), 0, ld, None, None, 0, loss.data_ptr(), pred.data_ptr(), None, B, C, S()) == -4
    assert lib.apgd_loss_pred(buf.data_ptr(), 9, ld, y.data_ptr(), None, 0, loss.data_ptr(), pred.data_ptr(), None, B, C, S()) == -3
    assert lib.apgd_loss_pred(buf.data_ptr(), 0, 4, y.data_ptr(), None, 0, loss.data_ptr(), pred.data_ptr(), None, B, C, S()) == -2


# --------------------------------------------------------------------------- K3 rows
@pytest.mark.parametrize("E,gelt", [(48, 4), (3 * 32 * 32, 4), (50, 4), (3 * 32 * 32, 2), (6, 2)])
@pytest.mark.parametrize("final", [0, 1])
def test_track_rows_semantics(lib, E, gelt, final):
    B = 16
    rng = np.random.default_rng(E + final)
    flags = (np.arange(B) % 8).astype(np.uint8)
    rng.shuffle(flags)
    gdt = torch.float32 if gelt == 4 else torch.bfloat16
    xa, xb, xba = (torch.rand(B, E, device="cuda") for _ in range(3))
    g, gb = (torch.randn(B, E, device="cuda").to(gdt) for _ in range(2))
    w = [t.clone() for t in (xa, g, xb, gb, xba)]
    for b in range(B):                                   # reference order: :304, :322-323, :345-346
        f = int(flags[b])
        if f & 2:
            w[4][b] = w[0][b]
        if f & 1:
            w[2][b] = w[0][b]
            if not final:
                w[3][b] = w[1][b]
        if f & 4 and not final:
            w[0][b] = w[2][b]
            w[1][b] = w[3][b]
    fd = dev(flags)
    assert lib.apgd_track_rows(fd.data_ptr(), xa.data_ptr(), g.data_ptr(), xb.data_ptr(), gb.data_ptr(),
                               xba.data_ptr(), gelt, B, E, final, S()) == 0
    for got, want in zip((xa, g, xb, gb, xba), w):
        assert torch.equal(got, want)


# --------------------------------------------------------------------------- end to end: replayed trajectories
class _ScriptedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, logits, grad):
        ctx.g = grad
        return logits.clone()

    @staticmethod
    def backward(ctx, gout):
        return ctx.g.clone(), None, None


class ReplayModule(torch.nn.Module):
    """Device twin of oracle.ReplayModel: returns the recorded logits / input gradients and
    remembers the sha256 of every iterate it is handed."""

    def __init__(self, g, channels_last=False):
        super().__init__()
        self.logits = [dev(l) for l in g["logits"]]
        self.grads = [dev(t) for t in g["grads"]]
        if channels_last:
            self.grads = [t.contiguous(memory_format=torch.channels_last) for t in self.grads]
        self.n, self.seen = 0, []

    def forward(self, x):
        n = self.n
        self.n += 1
        self.seen.append(sha(x))
        if x.requires_grad:
            return _ScriptedFn.apply(x, self.logits[n], self.grads[n])
        return self.logits[n].clone()


def _inject_losses(monkeypatch, R, losses):
    """Replace the loss the K2 kernel wrote by the reference's own fp32 value (keeps pred/dlogits):
    isolates K1/K3/state machine/host loop so EVERY golden case must be bit-exact."""
    real = R.apgd._loss_pred
    calls = {"n": 0}

    def patched(logits, y_hard, y_soft, loss_out, pred_out, want, kind=0):
        dl = real(logits, y_hard, y_soft, loss_out, pred_out, want, kind)
        loss_out.copy_(dev(losses[calls["n"]]))
        calls["n"] += 1
        return dl
    monkeypatch.setattr(R.apgd, "_loss_pred", patched)


def _run_replay(R, g, norm):
    x = dev(g["x"])
    if g["channels_last"]:
        x = x.contiguous(memory_format=torch.channels_last)
    y = dev(g["y"])
    m = ReplayModule(g, g["channels_last"]).eval()
    out = R.apgd_train(m, x, y, norm=norm, eps=g["eps"], n_iter=g["n_iter"], mixup=object() if g["soft"] else None,
                       loss=g.get("loss", "ce"))
    torch.cuda.synchronize()
    return x, m, out


@pytest.mark.parametrize("fused", [True, False], ids=["fused-track", "two-pass"])
@pytest.mark.parametrize("case", LINF)
def test_apgd_linf_golden_bit_exact_with_reference_losses(R, monkeypatch, case, fused):
    """``fused``: the row moves inside the next iteration's update kernel (apgd_linf_step_track_f32, the default since round 5)
    or as the separate apgd_track_rows pass - the same trajectories bit for bit either way."""
    monkeypatch.setattr(R.apgd, "FUSED_TRACKING", fused)
    g = load_golden(case)
    _inject_losses(monkeypatch, R, g["losses"])
    x, m, (xb, acc, lb, xba) = _run_replay(R, g, "Linf")
    assert m.seen == list(g["x_adv_sha"])                 # every iterate fed to the model, bit for bit
    assert bits_equal(xb.cpu().numpy(), g["x_best"])
    assert bits_equal(xba.cpu().numpy(), g["x_best_adv"])
    assert np.array_equal(acc.cpu().numpy(), g["acc"]) and acc.dtype == torch.bool
    assert bits_equal(lb.cpu().numpy(), g["loss_best"])
    assert xb.stride() == x.stride() and not xb.requires_grad    # memory format preserved


@pytest.mark.parametrize("case", TIE_FREE)
def test_apgd_linf_golden_full_hip_path(R, case):
    """No injection: the loss comes from the HIP kernel (tie-free cases, see test_oracle_golden.py)."""
    g = load_golden(case)
    x, m, (xb, acc, lb, xba) = _run_replay(R, g, "Linf")
    assert m.seen == list(g["x_adv_sha"])
    assert bits_equal(xb.cpu().numpy(), g["x_best"]) and bits_equal(xba.cpu().numpy(), g["x_best_adv"])
    assert np.array_equal(acc.cpu().numpy(), g["acc"])
    np.testing.assert_allclose(lb.cpu().numpy(), g["loss_best"], rtol=1e-5, atol=3e-7)


@pytest.mark.parametrize("case", L2)
def test_apgd_l2_golden(R, monkeypatch, case):
    g = load_golden(case)
    _inject_losses(monkeypatch, R, g["losses"])
    x, m, (xb, acc, lb, xba) = _run_replay(R, g, "L2")
    np.testing.assert_allclose(xb.cpu().numpy(), g["x_best"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xba.cpu().numpy(), g["x_best_adv"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(acc.cpu().numpy(), g["acc"])
    assert bits_equal(lb.cpu().numpy(), g["loss_best"])
    assert O.check_imgs(xb.cpu().numpy(), g["x"], "L2", g["eps"])[0] <= g["eps"] * (1 + 1e-5)


@pytest.mark.parametrize("B,E", [(3, 50), (4, 3 * 8 * 8), (2, 1021)])
def test_l2_step_vs_oracle(lib, B, E):
    eps = 0.5
    x, xa, xo, g, step = _step_inputs(B, E, 11, 0.05)
    g = np.nan_to_num(g)
    step = np.full(B, 1.0, np.float32)
    want = O.l2_step(x, xa, xo, g, step, eps, 0.75)
    xd, xad, xod, gd, sd = map(dev, (x, xa, xo, g, step))
    out = torch.empty_like(xd)
    ws = torch.empty(3 * B * lib.apgd_l2_parts(), device="cuda")
    assert lib.apgd_l2_step_f32(xd.data_ptr(), xad.data_ptr(), xod.data_ptr(), gd.data_ptr(), sd.data_ptr(),
                                out.data_ptr(), ws.data_ptr(), B, E, eps, 0.75, S()) == 0
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fused", [True, False], ids=["fused-track", "two-pass"])
@pytest.mark.parametrize("K", [2, 7, 40])
def test_apgd_fuzz_against_oracle(R, monkeypatch, K, fused):
    monkeypatch.setattr(R.apgd, "FUSED_TRACKING", fused)
    """Random scripted trajectories (many ties, halvings, mis-classifications): the oracle is the
    yardstick, driven by the same logits/grads/losses."""
    rng = np.random.default_rng(K)
    B, C, shape = 9, 6, (3, 5, 5)
    x = rng.random((B,) + shape, dtype=np.float32)
    y = rng.integers(0, C, B)
    logits = (rng.standard_normal((K + 1, B, C)) * 2).astype(np.float32)
    for n in range(2, K + 1, 3):
        logits[n, ::2] = logits[n - 1, ::2]                     # exact loss ties
    grads = rng.standard_normal((K,) + x.shape).astype(np.float32)
    grads[rng.random(grads.shape) < 0.05] = 0
    losses = np.stack([O.ce_loss(l, y) for l in logits])
    g = dict(x=x, y=y, logits=logits, grads=grads, losses=losses, eps=8 / 255, n_iter=K, soft=False,
             channels_last=False)
    rep = O.ReplayModel(logits, grads, losses)
    oxb, oacc, olb, oxba, tr = O.apgd_train_oracle(rep, x, y, "Linf", g["eps"], K, use_model_loss=True)
    _inject_losses(monkeypatch, R, losses)
    _, m, (xb, acc, lb, xba) = _run_replay(R, g, "Linf")
    assert m.seen == rep.seen_sha
    assert bits_equal(xb.cpu().numpy(), oxb) and bits_equal(xba.cpu().numpy(), oxba)
    assert np.array_equal(acc.cpu().numpy(), oacc) and bits_equal(lb.cpu().numpy(), olb)
    if K >= 7:
        assert tr.n_halvings > 0


# --------------------------------------------------------------------------- live model, boundary, errors
class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.fc = torch.nn.Linear(8, 10)

    def forward(self, x):
        return self.fc(torch.nn.functional.gelu(self.c1(x)).mean((-2, -1))) * 4


def test_apgd_live_model_matches_oracle_statistically(R):
    """Same weights on the device (HIP attack) and on the CPU (oracle attack + torch CPU model).
    GPU/CPU convolutions round differently, so a few gradient signs may flip: require the
    invariants exactly and agreement of the perturbation on >= 99 % of the pixels."""
    torch.manual_seed(0)
    m = _Tiny().eval()
    x = torch.rand(16, 3, 16, 16)
    y = torch.randint(0, 10, (16,))
    eps, K = 4 / 255, 3
    oxb, oacc, olb, oxba, _ = O.apgd_train_oracle(O.TorchModelAdapter(m, y.numpy()), x.numpy(), y.numpy(), "Linf", eps, K)
    xb, acc, lb, xba = R.apgd_train(m.cuda(), x.cuda(), y.cuda(), norm="Linf", eps=eps, n_iter=K)
    mx, n_nan, lo, hi = O.check_imgs(xb.cpu().numpy(), x.numpy(), "Linf", eps)
    assert n_nan == 0 and lo >= 0 and hi <= 1 and mx <= eps * (1 + 1e-6) + 1e-7
    assert (xb.cpu().numpy() == oxb).mean() >= 0.99
    assert np.array_equal(acc.cpu().numpy(), oacc)
    np.testing.assert_allclose(lb.cpu().numpy(), olb, rtol=2e-3, atol=1e-4)


def test_wrapped_model_boundary_and_autocast(R):
    torch.manual_seed(1)
    base = _Tiny().cuda()
    wm = R.wrap_model_for_at(base, R.AdvConfig.from_argv("--adv.attack apgd --adv.n_iter 2 --adv.eps 4/255".split()))
    assert isinstance(wm, R.WrappedModel) and all(k.startswith("base_model.") for k in wm.state_dict())
    x = torch.rand(8, 3, 16, 16, device="cuda")
    y = torch.randint(0, 10, (8,), device="cuda")
    wm.train()
    clean = wm(x)                                   # perturb off -> clean forward (main.py:295-298)
    assert torch.equal(clean, base(x))
    wm.set_perturb(True)
    with pytest.raises(AssertionError):
        wm(x)                                       # y is required (main.py:276)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = wm(x, y)
    assert out.dtype == torch.bfloat16 and base.training          # train mode restored (main.py:289)
    out.float().sum().backward()
    assert all(p.grad is not None for p in base.parameters())
    # mixup-style soft labels through the same boundary
    ys = torch.softmax(torch.randn(8, 10, device="cuda"), 1)
    xb, acc, lb, xba = R.apgd_train(base.eval(), x, ys, norm="Linf", eps=4 / 255, n_iter=2, mixup=object())
    assert acc.dtype == torch.bool and xb.shape == x.shape and float((xb - x).abs().max()) <= 4 / 255 + 1e-7


def test_error_behaviour(R):
    m = _Tiny().cuda()
    x = torch.rand(2, 3, 8, 8, device="cuda")
    y = torch.zeros(2, dtype=torch.long, device="cuda")
    with pytest.raises(AssertionError):
        R.apgd_train(m.train(), x, y, norm="Linf", eps=0.1, n_iter=1)        # autopgd_train_clean.py:125
    m.eval()
    with pytest.raises(KeyError):
        R.apgd_train(m, x, y, norm="Linf", eps=0.1, n_iter=1, loss="nope")   # :149
    with pytest.raises(Exception):
        R.apgd_train(m, x, y, norm="Linf", eps=0.1, n_iter=1, use_rs=True)   # :137
    with pytest.raises(R._lib.ApgdHipError):
        R.apgd_train(m, x.cpu(), y.cpu(), norm="Linf", eps=0.1, n_iter=1)    # no CPU fallback
    with pytest.raises(NotImplementedError):
        R.apgd_train(m, x, y, norm="L0", eps=0.1, n_iter=1)                  # L0 calls an undefined L0_projection in the reference (:257)


# ------------------------------------------------------------------------------ evaluation attacks (AA_eval.py path)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dlr_targeted_kernel_vs_reference_vectors(R, lib, dtype):
    """apgd_loss_pred_targeted vs the reference's own dlr_loss_targeted values / autograd gradients
    (tests/golden/loss_dlr_vectors.npz): fp32 bit-exact loss, gradient to 1e-5; bf16 logits vs the oracle on the
    rounded logits."""
    import os
    from conftest import ROOT
    v = np.load(os.path.join(ROOT, "tests", "golden", "loss_dlr_vectors.npz"))
    for tag in "abc":
        z, y, yt = v[f"{tag}_z"], v[f"{tag}_y"], v[f"{tag}_yt"]
        B, C = z.shape
        zt = dev(z).to(dtype)
        loss = torch.empty(B, device="cuda")
        pred = torch.empty(B, device="cuda", dtype=torch.uint8)
        dl = torch.empty_like(zt)
        assert lib.apgd_loss_pred_targeted(zt.data_ptr(), R._lib.dtype_code(dtype), C, dev(y).data_ptr(), dev(yt).data_ptr(),
                                           loss.data_ptr(), pred.data_ptr(), dl.data_ptr(), B, C, S()) == 0
        zf = zt.float().cpu().numpy()
        assert bits_equal(loss.cpu().numpy(), O.dlr_loss_targeted(zf, y, yt))
        assert np.array_equal(pred.cpu().numpy().astype(bool), O.predict(zf, y))
        if dtype == torch.float32:
            assert bits_equal(loss.cpu().numpy(), v[f"{tag}_dlr_t"])
            g = v[f"{tag}_dlr_t_grad"]
            np.testing.assert_allclose(dl.cpu().numpy(), g, rtol=1e-5, atol=1e-5 * float(np.abs(g).max()))
    assert lib.apgd_loss_pred_targeted(zt.data_ptr(), 0, 3, dev(y).data_ptr(), dev(yt).data_ptr(), loss.data_ptr(),
                                       pred.data_ptr(), None, 1, 3, S()) == -4            # needs >= 4 classes


def test_apgd_attack_targeted_random_start_matches_oracle(R):
    """aa_eval.apgd_attack (random start, targeted DLR, 12 iterations) vs the numpy oracle fed the SAME start point and
    the logits / input gradients the device model produced: bit-exact iterates and outputs."""
    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 8 * 8, 32), torch.nn.Tanh(), torch.nn.Linear(32, 10)).cuda().eval()
    B, K, eps = 9, 12, 8 / 255
    x = torch.rand(B, 3, 8, 8, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")
    with torch.no_grad():
        order = m(x).argsort(dim=1, descending=True)
    yt = torch.where(order[:, 1] == y, order[:, 0], order[:, 1])
    rec = {"logits": [], "grads": []}

    class Tap(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            rec["grads"].append(g.detach().float().cpu().numpy())
            return g

    class Rec(torch.nn.Module):
        def forward(self, t):
            out = m(Tap.apply(t) if t.requires_grad else t)
            rec["logits"].append(out.detach().float().cpu().numpy())
            return out

    gen = torch.Generator(device="cuda").manual_seed(11)
    x0 = R.aa_eval.random_start(x, eps, "Linf", torch.Generator(device="cuda").manual_seed(11))
    assert float((x0 - x).abs().amax()) <= eps * (1 + 1e-6)
    assert torch.allclose((x0 - x).reshape(B, -1).abs().amax(1), torch.full((B,), eps, device="cuda"), rtol=1e-5)
    xba, acc, lb, xb = R.aa_eval.apgd_attack(Rec().eval(), x, y, "Linf", eps, K, "dlr-targeted", yt, True, gen)
    torch.cuda.synchronize()
    rep = O.ReplayModel(np.stack(rec["logits"]), np.stack(rec["grads"]))
    oxb, oacc, olb, oxba, _ = O.apgd_train_oracle(rep, x.cpu().numpy(), y.cpu().numpy(), "Linf", eps, K, loss="dlr-targeted",
                                                  y_target=yt.cpu().numpy(), x_init=x0.cpu().numpy())
    assert bits_equal(xb.cpu().numpy(), oxb) and bits_equal(xba.cpu().numpy(), oxba)
    assert np.array_equal(acc.cpu().numpy(), oacc) and bits_equal(lb.cpu().numpy(), olb)
    with pytest.raises(ValueError):
        R.aa_eval.apgd_attack(m, x, y, loss="dlr-targeted")
    with pytest.raises(KeyError):
        R.aa_eval.apgd_attack(m, x, y, loss="nope")


def test_run_standard_evaluation_invariants_and_sharding(R):
    """AA_eval.py's caller shape: host tensors in, APGD-CE + APGD-T on still-robust points only, adversarials inside the
    ball, flags consistent with the model's predictions, shards rank::world cover the set exactly once."""
    torch.manual_seed(2)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, 2, 1), torch.nn.GELU(), torch.nn.Flatten(),
                            torch.nn.Linear(8 * 8 * 8, 12)).cuda().eval()
    n, eps = 26, 8 / 255
    x = torch.rand(n, 3, 16, 16)
    with torch.no_grad():
        y = m(x.cuda()).argmax(1).cpu()
    y[::5] = (y[::5] + 1) % 12                                    # some clean errors
    tot = {"n": 0, "clean_correct": 0, "robust": 0}
    for rank in range(2):
        x_adv, st = R.aa_eval.run_standard_evaluation(m, x, y, bs=7, eps=eps, n_iter=10, n_target_classes=3, seed=1, rank=rank,
                                                      world=2, device="cuda")
        xs, ys = x[rank::2], y[rank::2]
        assert x_adv.shape == xs.shape and st["n"] == xs.shape[0]
        assert float((x_adv - xs).abs().max()) <= eps * (1 + 1e-6) + 1e-7 and float(x_adv.min()) >= 0 and float(x_adv.max()) <= 1
        with torch.no_grad():
            clean_ok = (m(xs.cuda()).argmax(1).cpu() == ys)
            adv_ok = (m(x_adv.cuda()).argmax(1).cpu() == ys)
        assert st["clean_correct"] == int(clean_ok.sum())
        assert st["robust"] == int((clean_ok & adv_ok).sum())           # broken points carry a misclassified iterate
        assert torch.equal(x_adv[~clean_ok], xs[~clean_ok])             # never attacked
        for k in tot:
            tot[k] += st[k]
    assert tot["n"] == n and tot["robust"] <= tot["clean_correct"] <= n
    ca, ra = R.aa_eval.robust_accuracy(tot)
    assert abs(ca - tot["clean_correct"] / n) < 1e-12 and abs(ra - tot["robust"] / n) < 1e-12
    with pytest.raises(NotImplementedError):
        R.aa_eval.run_standard_evaluation(m, x, y, attacks_to_run=("square",), device="cuda")


def test_run_standard_evaluation_with_padded_batches_and_graph_replay(R):
    """``buckets`` / ``graph`` (round 5, BASELINE config #5): the still-robust subsets padded to a few fixed batch sizes with rows
    that are never read back, the attack runs replayed from hipGraphs once a shape has recurred - the evaluation's counts are
    those of the plain loop, the adversarials agree (bit for bit wherever the model's kernels do not depend on the batch size;
    the library convolution of this toy model may pick another algorithm for another batch, so >= 99.5 % identical values is the
    bar), padding rows leak nowhere, and programs really were captured and replayed."""
    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, 2, 1), torch.nn.GELU(), torch.nn.Flatten(),
                            torch.nn.Linear(8 * 8 * 8, 12)).cuda().eval()
    n, eps, bs = 96, 6 / 255, 32
    x = torch.rand(n, 3, 16, 16)
    with torch.no_grad():
        y = m(x.cuda()).argmax(1).cpu()
    y[::7] = (y[::7] + 1) % 12
    R.graphed.reset()
    before = dict(R.graphed.STATS)
    xa0, st0 = R.aa_eval.run_standard_evaluation(m, x, y, bs=bs, eps=eps, n_iter=8, n_target_classes=3, seed=3, device="cuda")
    xa1, st1 = R.aa_eval.run_standard_evaluation(m, x, y, bs=bs, eps=eps, n_iter=8, n_target_classes=3, seed=3, device="cuda",
                                                 buckets=True)
    xa2, st2 = R.aa_eval.run_standard_evaluation(m, x, y, bs=bs, eps=eps, n_iter=8, n_target_classes=3, seed=3, device="cuda",
                                                 buckets=True, graph=True)
    for xa, st in ((xa1, st1), (xa2, st2)):
        assert st["n"] == st0["n"] and st["clean_correct"] == st0["clean_correct"] and st["attack_runs"] == st0["attack_runs"]
        assert st["sample_iters"] == st0["sample_iters"]                      # padding rows are not work
        assert abs(st["robust"] - st0["robust"]) <= 1
        assert xa.shape == x.shape and float((xa - x).abs().max()) <= eps * (1 + 1e-6) + 1e-7
        assert float((xa == xa0).float().mean()) >= 0.995
    assert R.graphed.STATS["captures"] > before["captures"] and R.graphed.STATS["replays"] > before["replays"]
    assert R.graphed.STATS["failed"] == before["failed"]
    R.graphed.reset()
    assert [R.aa_eval._bucket(k, 100) for k in (1, 13, 14, 25, 26, 51, 100)] == [13, 13, 25, 25, 50, 100, 100]
