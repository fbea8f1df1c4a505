"""GPU parity tests at the BASELINE.json configurations, product (HIP path) vs the PINNED oracle on the host's CPU in fp32
(``oracle/models_ref.py`` restates the reference's models and is pinned to fixtures from the reference's own classes;
``oracle/apgd_oracle.py`` restates ``apgd_train`` bit-exactly).  One test per configuration the round-1 verdict listed as
never exercised against the oracle, plus the whole adversarial-training step and the (f) rows.

  cfg #1  ConvNeXt-iso-CvSt, APGD-2, 64x64, batch 4            models/convnext_iso.py:19-66, utils_architecture.py:235-239
  cfg #2  ConvNeXt-T-CvSt (full widths 96/192/384/768), 224x224 utils_architecture.py:241-244
  cfg #3  ViT-S / ViT-B-CvSt vs ViTTimm, 224x224                utils_architecture.py:271-301 (timm body: parity unpinned)
  cfg #4  ConvNeXt-L-CvSt, APGD-3, 320x320                      utils_architecture.py:264-269
  a11     3 full AT steps vs a CPU oracle step                  main.py:961-997, 395-459, 882-887

Tolerances (round 3: every bar at <= ~3x the value measured on MI355X, gpurun_out/test_metrics.jsonl - round 2's were 10-100x
looser than what the kernels deliver, wide enough for a dropped residual gradient to pass).  fp32 (no autocast): logits <= 5e-6,
input gradient <= 6e-5 relative L2 (measured 5e-7 ... 1.5e-6 / 1.2e-5 ... 1.8e-5), attack loss <= 5e-7 (6e-8 ... 1.4e-7), >= 99.99 %
identical pixels.  bf16 autocast: the error of the
hand-written path against the fp32 oracle must not exceed max(1e-2, 1.5 x the error of the plain bf16 library composition of
the same weights) - i.e. north_star's 1e-2, relaxed only where bf16 itself (not our kernels) cannot do better over a
60-kernel-deep chain; the measured values are appended to gpurun_out/test_metrics.jsonl.  Adversarial images: the attack
consumes sign(gradient), so end-to-end pixels agree except where the gradient is within rounding noise of zero; the state
machine itself is checked bit-exactly by replaying the device model's own logits / gradients through the oracle.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT
from oracle import apgd_oracle as O
from oracle import models_ref as M

pytestmark = pytest.mark.gpu
EPS = 4 / 255


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R
    assert torch.cuda.is_available()
    R._lib.load()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))       # the CPU oracle; oversubscription is far slower
    return R


def note(test, **kw):
    """Measured error levels, for the builder's records (never asserted on)."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_metrics.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if not isinstance(v, (str, list, dict)) else v)
                                                   for k, v in kw.items()})) + "\n")
    except OSError:
        pass


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def randomize(m, seed=0):
    """Layer scale at 0.5 (its 1e-6 init would hide every block branch), non-trivial LayerNorm weights and biases."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.5)
            elif p.ndim == 1 and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m


def build_pair(R, arch, res=224, seed=0):
    torch.manual_seed(seed)
    ref = randomize(M.build(arch, not_original=True, img_size=res), seed).eval()
    prod = R.get_new_model(arch, pretrained=False, not_original=True, img_size=res)
    prod.load_state_dict(ref.state_dict(), strict=True)               # identical keys: the checkpoint contract
    prod = prod.cuda().to(memory_format=torch.channels_last).eval()
    return ref, prod


def oracle_fwd_grad(ref, x, y):
    xr = x.clone().requires_grad_()
    logits = ref(xr)
    (g,) = torch.autograd.grad(F.cross_entropy(logits, y, reduction="sum"), xr)
    return logits.detach(), g


def product_fwd_grad(R, prod, x, y, autocast, mode="hip", monkeypatch=None):
    if monkeypatch is not None:
        monkeypatch.setattr(R.ops, "MODE", mode)
    R.ops.invalidate_weight_cache()
    torch.clear_autocast_cache()
    xd = x.cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logits = prod(xd)
    (g,) = torch.autograd.grad(F.cross_entropy(logits.float(), y.cuda(), reduction="sum"), xd)
    return logits.detach().float().cpu(), g.detach().float().cpu()


FP32_BARS = (5e-6, 6e-5)            # relative L2 of logits / input gradient vs the pinned oracle, fp32


def check_model_parity(R, monkeypatch, name, arch, res, nb, fp32_bars=FP32_BARS):
    ref, prod = build_pair(R, arch, res)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(nb, 3, res, res, generator=g)
    y = torch.randint(0, 1000, (nb,), generator=g)
    lo_r, gx_r = oracle_fwd_grad(ref, x, y)
    lo_f, gx_f = product_fwd_grad(R, prod, x, y, autocast=False, mode="hip", monkeypatch=monkeypatch)
    lo_h, gx_h = product_fwd_grad(R, prod, x, y, autocast=True, mode="hip", monkeypatch=monkeypatch)
    lo_e, gx_e = product_fwd_grad(R, prod, x, y, autocast=True, mode="eager", monkeypatch=monkeypatch)
    m = dict(fp32_logits=rel(lo_f, lo_r), fp32_grad=rel(gx_f, gx_r), bf16_logits=rel(lo_h, lo_r), bf16_grad=rel(gx_h, gx_r),
             bf16_lib_logits=rel(lo_e, lo_r), bf16_lib_grad=rel(gx_e, gx_r))
    note(name, arch=arch, res=res, **m)
    assert torch.equal(lo_f.argmax(1), lo_r.argmax(1))                          # class indices
    assert m["fp32_logits"] <= fp32_bars[0] and m["fp32_grad"] <= fp32_bars[1], m
    assert m["bf16_logits"] <= max(1e-2, 1.5 * m["bf16_lib_logits"]), m
    assert m["bf16_grad"] <= max(1e-2, 1.5 * m["bf16_lib_grad"]), m
    return ref, prod, x, y


def attack_both(R, ref, prod, x, y, K, autocast, norm="Linf", eps=EPS):
    oxb, oacc, olb, oxba, _ = O.apgd_train_oracle(O.TorchModelAdapter(ref, y.numpy()), x.numpy(), y.numpy(), norm, eps, K)
    R.ops.invalidate_weight_cache()
    torch.clear_autocast_cache()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        xb, acc, lb, xba = R.apgd_train(prod, x.cuda(), y.cuda(), norm=norm, eps=eps, n_iter=K)
    torch.cuda.synchronize()
    xb = xb.cpu().numpy()
    same = float((xb == oxb).mean())
    moved = float((oxb != np.clip(x.numpy(), 0, 1)).mean())
    assert float(np.abs(xb - x.numpy()).max()) <= eps * (1 + 1e-6) + 1e-7 and xb.min() >= 0 and xb.max() <= 1
    return dict(same=same, moved=moved, acc_equal=bool(np.array_equal(acc.cpu().numpy(), oacc)),
                loss_rel=float(np.abs(lb.cpu().numpy() - olb).max() / (np.abs(olb).max() + 1e-30)))


def replay_is_bit_exact(R, prod, x, y, K, autocast, agree_bar=0.99):
    """The device model's own logits / input gradients, recorded during the HIP attack and replayed through the numpy
    oracle, must give the HIP attack's outputs bit for bit (as __graft_entry__.smoke does).  Checked in both modes: fp32
    gradients through autograd, and the int8 gradient-sign sink the product path uses by default (the recorded 'gradient' is
    then the sign tensor the stem kernel wrote).  The two runs are only required to AGREE on >= 99 % of the pixels: library
    backward kernels are not run-to-run reproducible (oracle/replay_tap.py), and one flipped sign moves a pixel by a step."""
    from oracle import replay_tap as T
    xd, yd = x.cuda(), y.cuda()
    out_f32, lo, gr, _ = T.record_attack(R, prod, xd, yd, "Linf", EPS, K, autocast=autocast, sink=False)
    T.check_replay(O, out_f32, lo, gr, x, y, "Linf", EPS, K)
    out_i8, lo, gr, used = T.record_attack(R, prod, xd, yd, "Linf", EPS, K, autocast=autocast, sink=True)
    assert used and all(used), "the product model did not hand the attack its int8 gradient signs"
    assert set(np.unique(gr)) <= {-1.0, 0.0, 1.0}
    T.check_replay(O, out_i8, lo, gr, x, y, "Linf", EPS, K)
    agree = float((out_f32[0] == out_i8[0]).float().mean())
    note("sink_vs_fp32_gradient_runs", agree=agree)
    assert agree >= agree_bar, agree


# ------------------------------------------------------------------------------------------------ cfg #2 (full widths)
def test_cfg2_convnext_tiny_cvst_224_vs_oracle(R, monkeypatch):
    ref, prod, x, y = check_model_parity(R, monkeypatch, "cfg2_model", "convnext_tiny", 224, 2)
    monkeypatch.setattr(R.ops, "MODE", "hip")
    a32 = attack_both(R, ref, prod, x, y, 2, autocast=False)
    a16 = attack_both(R, ref, prod, x, y, 2, autocast=True)
    note("cfg2_attack", fp32=a32, bf16=a16)
    assert a32["same"] >= 0.9999 and a32["acc_equal"] and a32["loss_rel"] <= 5e-7, a32
    assert a16["same"] >= 0.95 and a16["acc_equal"] and a16["loss_rel"] <= 3e-3, a16          # measured 0.970 / 7e-4
    replay_is_bit_exact(R, prod, x, y, 2, autocast=True)


@pytest.mark.parametrize("nb", [3, 5])
def test_cfg2_convnext_tiny_cvst_224_ragged_batches(R, monkeypatch, nb):
    """The same configuration at batch sizes that leave ragged tiles everywhere (3 and 5 images: 9408 / 15680 rows at 56x56 are
    not multiples of the 128-row workgroups, 147 / 245 rows at 7x7 not even of a wavefront's 32): model parity at the fp32 and
    bf16 bars, and the bf16 attack replayed bit for bit through the oracle in both gradient modes."""
    ref, prod, x, y = check_model_parity(R, monkeypatch, f"cfg2_model_b{nb}", "convnext_tiny", 224, nb)
    monkeypatch.setattr(R.ops, "MODE", "hip")
    a16 = attack_both(R, ref, prod, x, y, 2, autocast=True)
    note(f"cfg2_attack_b{nb}", bf16=a16)
    assert a16["same"] >= 0.95 and a16["loss_rel"] <= 3e-3, a16
    replay_is_bit_exact(R, prod, x, y, 2, autocast=True)


# ------------------------------------------------------------------------------------------------ cfg #4
def test_cfg4_convnext_large_cvst_320_apgd3_vs_oracle(R, monkeypatch):
    """80x80x192 / 40x40x384 / 20x20x768 / 10x10x1536 maps: stage 0 leaves the 56x56 rolling depthwise kernel's shape."""
    ref, prod, x, y = check_model_parity(R, monkeypatch, "cfg4_model", "convnext_large", 320, 2)
    monkeypatch.setattr(R.ops, "MODE", "hip")
    a16 = attack_both(R, ref, prod, x, y, 3, autocast=True)
    note("cfg4_attack", bf16=a16)
    assert a16["same"] >= 0.97 and a16["acc_equal"] and a16["loss_rel"] <= 3e-3, a16   # measured 0.993 / 6e-4 (three sign steps)
    replay_is_bit_exact(R, prod, x, y, 3, autocast=True)


# ------------------------------------------------------------------------------------------------ cfg #1
def test_cfg1_convnext_iso_cvst_64_apgd2_vs_oracle(R, monkeypatch):
    """BASELINE config #1 on the device: 4x4x384 feature maps under a 7x7 stencil, gamma-less blocks, bf16 residual."""
    ref, prod, x, y = check_model_parity(R, monkeypatch, "cfg1_model", "convnext_iso", 64, 4)
    monkeypatch.setattr(R.ops, "MODE", "hip")
    a32 = attack_both(R, ref, prod, x, y, 2, autocast=False)
    a16 = attack_both(R, ref, prod, x, y, 2, autocast=True)
    note("cfg1_attack", fp32=a32, bf16=a16)
    assert a32["same"] >= 0.9999 and a32["acc_equal"] and a32["loss_rel"] <= 5e-7, a32
    assert a16["same"] >= 0.95 and a16["acc_equal"] and a16["loss_rel"] <= 3e-3, a16          # measured 0.974 / 4e-4
    replay_is_bit_exact(R, prod, x, y, 2, autocast=True)
    # the reference's plumbing config runs L2 as well (secondary norm)
    l2 = attack_both(R, ref, prod, x, y, 2, autocast=False, norm="L2", eps=2.0)
    note("cfg1_attack_l2", fp32=l2)
    assert l2["acc_equal"] and l2["loss_rel"] <= 5e-7, l2      # (L2 iterates are real-valued: pixels agree to rounding, not bit for bit)


# ------------------------------------------------------------------------------------------------ cfg #3
@pytest.mark.parametrize("arch", ["vit_s", "vit_b", "vit_m"])
def test_cfg3_vit_cvst_vs_oracle_vittimm(R, monkeypatch, arch):
    """Product ViT (fused attention fwd/bwd, row LayerNorm, scale-residual kernels, ConvStem kernels) vs
    ``models_ref.ViTTimm`` with shared weights.  (The timm body is absent from the reference: ViTTimm is unpinned.)"""
    ref, prod, x, y = check_model_parity(R, monkeypatch, f"cfg3_{arch}", arch, 224, 2)
    if arch == "vit_b":
        monkeypatch.setattr(R.ops, "MODE", "hip")
        a16 = attack_both(R, ref, prod, x, y, 2, autocast=True)
        note("cfg3_attack", bf16=a16)
        assert a16["same"] >= 0.95 and a16["acc_equal"] and a16["loss_rel"] <= 3e-3, a16      # measured 0.967 ... 0.991 / 1.0e-3


def test_convnext_base_cvst_convblock3_vs_oracle(R, monkeypatch):
    """ConvNeXt-B-CvSt (ConvBlock3 stem, widths 128/256/512/1024: BASELINE config #5's model); the bf16 attack runs the
    C = 128 / 256 stages on the Hpre forward / input-gradient pair."""
    ref, prod, x, y = check_model_parity(R, monkeypatch, "convnext_base_model", "convnext_base", 224, 2)
    monkeypatch.setattr(R.ops, "MODE", "hip")
    assert R.ops._use_hpre_block(128) and R.ops._use_hpre_block(256)
    a16 = attack_both(R, ref, prod, x, y, 2, autocast=True)
    note("convnext_base_attack", bf16=a16)
    assert a16["same"] >= 0.97 and a16["acc_equal"] and a16["loss_rel"] <= 3e-3, a16          # measured 0.993 / 4e-4
    # (the ConvBlock3 stem's stride-1 convolution runs MIOpen's backward-data kernel, which is not reproducible run to run:
    # each run must replay bit for bit, two runs of two images may differ in a few per cent of the pixels)
    replay_is_bit_exact(R, prod, x, y, 2, autocast=True, agree_bar=0.90)


# ------------------------------------------------------------------------------------------------ reference fixtures on the HIP path
def _fixture(name):
    d = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w::")}
    return sd, torch.from_numpy(d["x"]), d["out"], d["gx"], torch.from_numpy(d["cot"])


@pytest.mark.parametrize("name", ["ln_cf", "cn_block", "stem_block1", "stem_block3", "stem_block", "stem_block2",
                                  "convnext_iso_cvst", "convnext_t_cvst", "normalize_model"])
def test_product_modules_on_the_hip_path_match_reference_fixtures(R, name):
    """Forward output and input gradient recorded from the REFERENCE's own classes (tests/golden/make_model_golden.py)
    vs the product classes running the hand-written fp32 kernels (no autocast): ConvBlock/1/2/3, LayerNorm(channels_first),
    the ConvNeXt block, ConvNeXtIsotropic, staged ConvNeXt, normalize_model."""
    from test_product_surface import product_builders
    assert R.ops.MODE == "hip"
    sd, x, out, gx, cot = _fixture(name)
    m = product_builders()[name]()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    xd = x.cuda().requires_grad_()
    y = m(xd)
    (g,) = torch.autograd.grad((y * cot.cuda()).sum(), xd)
    np.testing.assert_allclose(y.detach().cpu().numpy(), out, rtol=1e-4, atol=2e-5)
    assert rel(g, torch.from_numpy(gx)) <= 1e-3


@pytest.mark.parametrize("name", ["stem_block3", "stem_block1"])
def test_narrow_fp32_stems_survive_an_allocator_layout_fuzz(name):
    """The fp32 ConvStem fixtures run while random small allocations move their tensors around the caching allocator's segments
    (``tools/probe/fault_fuzz.py``, own process: a GPU memory fault aborts it).  Guards the library work-around in
    ``architecture._StemConv2`` - MIOpen's fp32 NHWC backward-data solver read past its operands at widths 8 -> 12 and took one
    full-suite run in three down with it (profiles/r02_miopen_nhwc_bwd_fault.md)."""
    import subprocess
    import sys
    last = os.path.join(ROOT, "gpurun_out", f"fuzz_last_test_{name}.txt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "fault_fuzz.py"), "--name", name, "--iters", "250",
                        "--no-sync", "--last", last], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean 250" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-600:])


# ------------------------------------------------------------------------------------------------ a11: whole AT step
def _oracle_at_steps(ref, x, y, n_steps, lr, decay=0.9999):
    """CPU fp32 restatement of main.py:961-997 with the reference's optimizer groups (main.py:395-459: names containing
    'bn' or '.bias' are not decayed for convnext archs), AdamW(0.9, 0.95) and ModelEmaV2(decay)."""
    named = list(ref.named_parameters())
    nd = [p for n, p in named if ("bn" in n or ".bias" in n)]
    dc = [p for n, p in named if not ("bn" in n or ".bias" in n)]
    opt = torch.optim.AdamW([{"params": nd, "weight_decay": 0.0}, {"params": dc, "weight_decay": 0.05}], lr=lr,
                            betas=(0.9, 0.95))
    ema = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    losses, grads1 = [], None
    for s in range(n_steps):
        ref.eval()
        xb = O.apgd_train_oracle(O.TorchModelAdapter(ref, y.numpy()), x.numpy(), y.numpy(), "Linf", EPS, 2)[0]
        ref.train()
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(ref(torch.from_numpy(xb)), y)
        loss.backward()
        if s == 0:
            grads1 = {n: p.grad.detach().clone() for n, p in named}
        opt.step()
        with torch.no_grad():
            for k, v in ref.state_dict().items():
                ema[k].mul_(decay).add_(v.detach(), alpha=1 - decay)
        losses.append(float(loss))
    return losses, grads1, ema


@pytest.mark.parametrize("mode", ["eager", "graph", "graph-flat"])
@pytest.mark.parametrize("amp", [None, torch.bfloat16], ids=["fp32", "bf16"])
def test_at_train_step_matches_oracle_step(R, amp, mode):
    _at_step_case(R, amp, mode, 64, 4)


@pytest.mark.parametrize("mode", ["graph", "graph-flat"])
def test_at_train_step_full_size_captured_pass_matches_oracle_step(R, mode):
    """The same comparison at the benchmark's resolution - ConvNeXt-T-CvSt at 224 x 224, batch 32, bf16 - for the two captured forms of
    the step (round-4 review: the captured TRAINING pass had its oracle check at 64 x 64, batch 4 only).  At this size every stage but
    the last has a row count the round-5 training pass takes (Hpre kernel pair at C = 192 / 384, accumulator-order emit at C = 96,
    cnx_gemm_tn weight gradients), the attack replays on two streams, and steps 4 - 5 run from the captured training pass."""
    _at_step_case(R, torch.bfloat16, mode, 224, 32)


def _at_step_case(R, amp, mode, res, batch):
    """``mode``: "eager" - every kernel launched from Python (rounds 1 - 3 checked only this form against the oracle); "graph" - the
    step ``bench.py`` times: attack replayed from hipGraphs (two streams under bf16), training pass captured with the capturable
    AdamW, five steps so that the fourth is the capture's first replay and the fifth a plain replay; "graph-flat" - the same with
    the gradients through ``FlatGradSync`` (the path of an N > 1 rank: two-call backward, three graph segments).

    Three full product steps (attack + train fwd/bwd + AdamW with the reference's groups + EMA) on ConvNeXt-T-CvSt at
    64x64, batch 4, against the CPU oracle step on the same seeds: loss trajectory, first-step parameter gradients,
    parameter updates and EMA.  Under bf16 the fused LN+MLP kernels (training backward with operand emission), split-K
    weight gradients, depthwise filter gradients, stem filter gradient and the weight-cache invalidation all run."""
    torch.manual_seed(3)
    ref = randomize(M.build("convnext_tiny", not_original=True), 3)
    prod = R.get_new_model("convnext_tiny", pretrained=False, not_original=True)
    prod.load_state_dict(ref.state_dict(), strict=True)
    p0 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.rand(batch, 3, res, res, generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)
    lr, n_steps = 1e-3, (3 if mode == "eager" else 5)
    R.graphed.reset()

    tr = R.ATTrainStep(prod, "convnext_tiny", R.AdvConfig(attack="apgd", norm="Linf", eps=EPS, n_iter=2, graph=int(mode != "eager")),
                       "cuda", lr=lr, amp_dtype=amp, ema=True, ema_decay=0.9999, grad_sync="flat" if mode == "graph-flat" else None)
    assert tr.graph_train == (mode != "eager")
    miss0 = R.ops.CACHE_STATS["miss"]
    losses, traj, emas, grads1, misses = [], [], [], None, []
    xd, yd = x.cuda(), y.cuda()
    for s in range(n_steps):
        losses.append(float(tr.step(xd, yd)))
        if s == 0:
            grads1 = {n[len("base_model."):]: p.grad.detach().float().cpu().clone() for n, p in tr.inner.named_parameters()}
        traj.append({k[len("base_model."):]: v.detach().float().cpu().clone() for k, v in tr.inner.state_dict().items()})
        emas.append({k[len("base_model."):]: v.detach().float().cpu().clone() for k, v in tr.ema.state_dict().items()})
        misses.append(R.ops.CACHE_STATS["miss"])
    if amp is not None:                                   # derived weight copies are rebuilt after every optimizer step
        assert misses[0] > miss0 and misses[1] > misses[0] and misses[2] > misses[1], (miss0, misses)
    if mode != "eager":                                   # steps 4 and 5 ran from the captured attack and the captured training pass
        assert [v.n_graphs for v in tr._tg.values() if v is not None] == [3 if mode == "graph-flat" else 1], tr._tg
        assert R.graphed.STATS["replays"] >= 3 and R.graphed.STATS["failed"] == 0

    o_losses, o_grads1, o_ema = _oracle_at_steps(ref, x, y, n_steps, lr)
    o_final = {k: v.detach() for k, v in ref.state_dict().items()}

    def flat(d, keys):
        return torch.cat([d[k].flatten().float() for k in keys])
    keys = [n for n, _ in ref.named_parameters()]
    g_rel = rel(flat(grads1, keys), flat(o_grads1, keys))
    upd_p = flat(traj[-1], keys) - flat(p0, keys)
    upd_o = flat(o_final, keys) - flat(p0, keys)
    cos = float(F.cosine_similarity(upd_p.double(), upd_o.double(), dim=0))
    ema_p = flat(emas[-1], keys) - flat(p0, keys)
    ema_o = flat(o_ema, keys) - flat(p0, keys)
    ema_cos = float(F.cosine_similarity(ema_p.double(), ema_o.double(), dim=0))
    loss_rel = max(abs(a - b) / abs(b) for a, b in zip(losses, o_losses))
    note("at_step", amp=str(amp), mode=mode, res=res, batch=batch, losses=losses, o_losses=o_losses, grad_rel=g_rel, update_cos=cos, ema_cos=ema_cos,
         upd_norm_ratio=float(upd_p.norm() / upd_o.norm()))
    # the product's EMA is exactly the ModelEmaV2 recursion over the product's own parameter trajectory
    d = 0.9999
    prev = p0
    for k_step in range(n_steps):
        for k in keys:
            expect = d * prev[k] + (1 - d) * traj[k_step][k]
            torch.testing.assert_close(emas[k_step][k], expect, rtol=1e-5, atol=1e-7)
        prev = emas[k_step]
    # bars at <= 3x the measured values (fp32: loss 1.2e-5, gradient 3.7e-5, cosines 0.999997 / 0.99997; bf16: 4.7e-3, 6.2e-3,
    # 0.9984 / 0.9976) - round 2 asserted 1e-3 / 5e-3 / 0.98 and 3e-2 / 1e-1 / 0.80
    # (the five-step runs of the graph modes carry two more optimizer steps of drift between the two implementations: loss bar x2)
    k = 1.0 if mode == "eager" else 2.0
    if amp is None:
        assert loss_rel <= 4e-5 * k and g_rel <= 1.2e-4 and cos >= 0.9999 and ema_cos >= 0.9999, (loss_rel, g_rel, cos, ema_cos)
    else:
        assert loss_rel <= 1.5e-2 * k and g_rel <= 2e-2 and cos >= 0.99 and ema_cos >= 0.99, (loss_rel, g_rel, cos, ema_cos)
    R.graphed.reset()
    assert 0.99 <= float(upd_p.norm() / upd_o.norm()) <= 1.01


# ------------------------------------------------------------------------------------------------ (f) rows on the device
def test_mixup_cutmix_targets_on_device_equal_host(R):
    """f2: the same seeded Mixup (main.py:599-607) on device tensors and on host tensors gives the same images and the same
    fp32 [B, n_cls] soft targets; K2 then consumes them (argmax of the soft label is the class compared with the prediction)."""
    from revisiting_at_amd.mixup import Mixup
    g = torch.Generator().manual_seed(0)
    x = torch.rand(8, 3, 32, 32, generator=g)
    y = torch.randint(0, 1000, (8,), generator=g)
    a = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000, seed=5)
    b = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000, seed=5)
    for _ in range(6):
        xh, yh = a(x, y)
        xd, yd = b(x.cuda(), y.cuda())
        assert xd.is_cuda and yd.is_cuda and yd.dtype == torch.float32 and yd.shape == (8, 1000)
        torch.testing.assert_close(xd.cpu(), xh, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(yd.cpu(), yh, rtol=1e-6, atol=1e-7)
    lib = R._lib.load()
    logits = torch.randn(8, 1000, device="cuda")
    loss = torch.empty(8, device="cuda")
    pred = torch.empty(8, device="cuda", dtype=torch.uint8)
    R.apgd._loss_pred(logits, None, yd.contiguous(), loss, pred, False)
    ref_loss = torch.sum(-yd * torch.log_softmax(logits, 1), 1)
    torch.testing.assert_close(loss, ref_loss, rtol=1e-5, atol=1e-5)
    assert torch.equal(pred.bool(), logits.argmax(1) == yd.argmax(1))


def test_pos_embed_interpolation_on_device_matches_reference_fixture(R):
    """f4: utils_architecture.py:22-53 on a device tensor (AA_eval.py:198-211 resizes the table for 320x320 evaluation)."""
    d = np.load(os.path.join(GOLDEN, "model_pos_embed.npz"))
    pe = torch.from_numpy(d["pos_embed"]).cuda()
    for res in (320, 256):
        got = R.architecture.interpolate_pos_encoding(pe, res, old_img_size=224, patch_size=16)
        np.testing.assert_allclose(got.cpu().numpy(), d[f"out_{res}"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ cfg #5, assembled
def _eps_with_survivors(AE, model, x, y, seed, K, lo_frac=0.35, hi_frac=0.65):
    """Largest-to-smallest bisection for an eps at which APGD-CE (the evaluation's own first run: same generator seed, same points)
    leaves between 35 % and 65 % of the points robust.  A random-init model has no robust point at 4/255."""
    lo, hi, eps = 0.0, EPS, EPS
    for _ in range(10):
        eps = 0.5 * (lo + hi)
        gen = torch.Generator(device="cuda").manual_seed(seed * 1000003)
        _, acc, _, _ = AE.apgd_attack(model, x, y, "Linf", eps, K, "ce", None, True, gen)
        frac = float(acc.float().mean())
        if lo_frac <= frac <= hi_frac:
            return eps, frac
        if frac > hi_frac:
            lo = eps
        else:
            hi = eps
    return eps, frac


def test_cfg5_standard_evaluation_on_convnext_base_replays_through_the_oracle(R, monkeypatch):
    """BASELINE config #5 as ``AA_eval.py:226-239`` drives it: ``run_standard_evaluation`` (APGD-CE, then APGD-T on what is still
    robust) on the ConvNeXt-B-CvSt PRODUCT model in fp32, 100 iterations, 2 target classes, 8 images at 224x224.  Every attack run
    of the evaluation is recorded (start point, logits, sign of the input gradient per model call) and replayed through the pinned
    numpy oracle: iterates, ``acc`` and ``loss_best`` must come out bit for bit, and the evaluation's bookkeeping (who is still
    robust, which adversarial is kept, the counts) must be what the oracle's results imply.

    Round 4: at eps = 4/255 APGD-CE breaks EVERY point of a random-init model, so rounds 2 - 3 only ever ran the CE leg here.  The
    evaluation now runs at an eps chosen (by bisection over its own first attack) so that about half of the points survive
    APGD-CE: both targeted runs execute on the product model - ``dlr-targeted``, target classes from the product's clean logits,
    the still-robust subset as their batch - and all THREE runs are replayed through the oracle."""
    from revisiting_at_amd import aa_eval as AE
    torch.manual_seed(0)
    model = randomize(R.get_new_model("convnext_base", pretrained=False, not_original=True), 0)
    model = model.cuda().to(memory_format=torch.channels_last).eval()
    n, K, eps = 8, 100, EPS
    g = torch.Generator().manual_seed(21)
    x = torch.rand(n, 3, 224, 224, generator=g)
    with torch.no_grad():
        y = model(x.cuda()).argmax(1).cpu()
    y[5] = (y[5] + 1) % 1000                                               # one clean error: never attacked
    idx0 = torch.tensor([i for i in range(n) if i != 5])
    eps, frac = _eps_with_survivors(AE, model, x[idx0].cuda().contiguous(), y[idx0].cuda(), 3, K)
    assert 2 / 7 <= frac <= 5 / 7, (eps, frac)

    calls = []
    orig_attack, orig_rs = AE.apgd_attack, AE.random_start

    def rs(xx, e, norm="Linf", generator=None):
        out = orig_rs(xx, e, norm, generator)
        calls[-1]["x_init"] = out.detach().cpu().numpy()
        return out

    def attack(m, xi, yi, norm, e, n_iter, loss, yt, use_rs, gen, graph=False, n_real=None):
        rec = {"x": xi.cpu().numpy(), "y": yi.cpu().numpy(), "loss": loss, "yt": None if yt is None else yt.cpu().numpy(),
               "logits": [], "signs": []}
        calls.append(rec)

        class Tap(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, gr):
                rec["signs"].append(torch.sign(gr).to(torch.int8).cpu().numpy())   # all the Linf update reads (:221)
                return gr

        class Rec(torch.nn.Module):
            def forward(self, t):
                o = m(Tap.apply(t) if t.requires_grad else t)
                rec["logits"].append(o.detach().float().cpu().numpy())
                return o
        out = orig_attack(Rec().eval(), xi, yi, norm, e, n_iter, loss, yt, use_rs, gen, False, n_real)   # (a recording model cannot be captured)
        rec["out"] = [t.cpu().numpy() for t in out]                          # x_best_adv, acc, loss_best, x_best
        return out

    # per-sample losses as the device computed them: at 100 iterations the iterates oscillate and their losses tie to the last
    # bits, where two correct cross-entropy implementations order them differently ("a note on ties") - the
    # oracle is driven by the device's own losses, as in the bit-exact golden tests
    orig_lp = R.apgd._loss_pred

    def loss_pred(logits, y_hard, y_soft, loss_out, pred_out, want_dl, kind=0, *a):
        dl = orig_lp(logits, y_hard, y_soft, loss_out, pred_out, want_dl, kind, *a)
        calls[-1].setdefault("losses", []).append(loss_out.clone())
        return dl

    monkeypatch.setattr(R.apgd, "_loss_pred", loss_pred)
    monkeypatch.setattr(AE, "apgd_attack", attack)
    monkeypatch.setattr(AE, "random_start", rs)
    monkeypatch.setattr(R.apgd, "USE_SIGN_SINK", False)                      # fp32 gradients through autograd: the tap sees them
    x_adv, st = AE.run_standard_evaluation(model, x, y, bs=8, eps=eps, n_iter=K, n_target_classes=2, seed=3, device="cuda")
    torch.cuda.synchronize()
    assert [c["loss"] for c in calls] == ["ce", "dlr-targeted", "dlr-targeted"], [c["loss"] for c in calls]
    assert st["attack_runs"] == 3
    # the targeted runs attack the survivors only: 2 - 5 of the 7 after APGD-CE, then whoever the first targeted run left robust
    assert 2 <= len(calls[1]["y"]) <= 5 and 1 <= len(calls[2]["y"]) <= len(calls[1]["y"]), [len(c["y"]) for c in calls]

    class Replay:
        def __init__(self, c):
            self.c, self.n = c, 0

        def __call__(self, x_adv_np, need_grad):
            i = self.n
            self.n += 1
            gsign = self.c["signs"][i].astype(np.float32) if need_grad else None
            return self.c["logits"][i], gsign, self.c["losses"][i].cpu().numpy()

    # the evaluation's bookkeeping, re-derived from the oracle's results
    with torch.no_grad():
        clean_logits = model(x.cuda()).float().cpu()
    robust = (clean_logits.argmax(1) == y).numpy()
    assert st["clean_correct"] == int(robust.sum()) == n - 1
    want_adv = x.numpy().copy()
    order = clean_logits.argsort(dim=1, descending=True).numpy()
    for ci, c in enumerate(calls):
        idx = np.nonzero(robust)[0]
        assert np.array_equal(c["x"], x.numpy()[idx]) and np.array_equal(c["y"], y.numpy()[idx])
        if c["loss"] == "dlr-targeted":
            assert np.array_equal(c["yt"], order[idx, ci])                   # ci = 1, 2 -> the 2nd / 3rd most likely class
        assert len(c["logits"]) == K + 1 and len(c["signs"]) == K
        assert len(c["losses"]) == K + 1
        oxb, oacc, olb, oxba, _ = O.apgd_train_oracle(Replay(c), c["x"], c["y"], "Linf", eps, K, loss=c["loss"], y_target=c["yt"],
                                                      x_init=c["x_init"], use_model_loss=True)
        xba, acc, lb, xb = c["out"]
        assert np.array_equal(xb, oxb) and np.array_equal(xba, oxba), f"attack {ci}: iterates differ from the oracle replay"
        assert np.array_equal(acc, oacc), f"attack {ci}"
        np.testing.assert_allclose(lb, olb, rtol=1e-5, atol=3e-7)
        assert float(np.abs(oxba - c["x"]).max()) <= eps * (1 + 1e-6) + 1e-7
        broken = ~oacc
        want_adv[idx[broken]] = oxba[broken]
        robust[idx[broken]] = False
    assert st["robust"] == int(robust.sum()) and st["n"] == n
    assert np.array_equal(x_adv.numpy(), want_adv)
    assert np.array_equal(x_adv.numpy()[5], x.numpy()[5])                    # the clean error was never touched
    note("cfg5_assembled", attacks=len(calls), robust=int(robust.sum()), clean_correct=int(st["clean_correct"]), eps=eps,
         survivors_after_ce=int(len(calls[1]["y"])), targeted_batches=[int(len(c["y"])) for c in calls[1:]],
         broken_by_targeted=[int((~c["out"][1]).sum()) for c in calls[1:]])


def test_at_train_step_is_bit_reproducible_run_to_run(R):
    """Two trainers from the same seed, the same four batches (two eager warm-up steps, two replayed ones): every parameter bit-identical
    afterwards.  Every partial sum of the step has a fixed order (cnx_gemm_tn's splits, the depthwise / LayerNorm / column-sum partials, the
    gradient identities' trees); nothing accumulates through atomics."""
    dev = torch.device("cuda")

    def run():
        torch.manual_seed(0)
        model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True, img_size=224)
        tr = R.ATTrainStep(model, "convnext_tiny", R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=2, graph=1), dev, lr=1e-3,
                           amp_dtype=torch.bfloat16, ema=True)
        g = torch.Generator(device=dev).manual_seed(7)
        losses = []
        for _ in range(4):
            x = torch.rand(16, 3, 224, 224, device=dev, generator=g)
            y = torch.randint(0, 1000, (16,), device=dev, generator=g)
            losses.append(float(tr.step(x, y)))
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in model.parameters()]

    la, pa = run()
    lb, pb = run()
    assert la == lb and all(np.isfinite(la))
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
