#!/usr/bin/env python3
"""Headline benchmark: adversarial images/sec of the APGD-2 adversarial-training step,
ConvNeXt-T-CvSt @224, bf16, per-GPU batch 256, synthetic data (BASELINE.json configs[1]).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one synthetic batch per GPU: APGD attack
(K+1 eval forwards, K input-gradient backwards, fused HIP update/track kernels) + train-mode
forward on x_best + backward (DDP/RCCL gradient all-reduce when N > 1) + AdamW + EMA.
Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).

`roofline` is for the hand-written APGD Linf update kernel (HBM-bound, 20 algorithmic bytes
per element per launch): its launches inside the timed region are bracketed with
HIP events on the stream they run on.  `cpu_baseline` times the CPU oracle (numpy attack +
plain-torch model) on a bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
K1_BYTES_PER_ELEM = 20.0       # read x, x_adv, x_adv_old, grad (16 B) + write x_adv (4 B)
K1_BYTES_PER_ELEM_IT0 = 16.0   # at i=0 x_adv_old aliases x_adv: one tensor fewer comes from HBM


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--arch", default="convnext_tiny")
    p.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    p.add_argument("--res", type=int, default=224)
    p.add_argument("--n-iter", type=int, default=2)
    p.add_argument("--eps", type=float, default=4 / 255)
    p.add_argument("--soft-labels", action="store_true", help="mixup-style [B,1000] targets")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-batch", type=int, default=16)
    p.add_argument("--cpu-steps", type=int, default=6)
    p.add_argument("--cpu-threads", type=int, default=32)
    p.add_argument("--attack-only", action="store_true", help="time only apgd_train (extra info line on stderr)")
    p.add_argument("--no-other-configs", action="store_true", help="skip the informational runs of BASELINE configs #3-#5")
    p.add_argument("--graph", type=int, default=1, help="1: the attack is replayed from hipGraphs (adv.graph, graphed.py); 0: eager")
    p.add_argument("--other-configs", default=None, help="which informational configurations (cfg3,cfg4,cfg5) to run after the timed "
                   "region; default: all three, none with --no-cpu-baseline / --no-other-configs")
    p.add_argument("--settle-steps", type=int, default=45, help="after the timed region: keep stepping until this many steps have run in "
                   "all (~2.4 s at the headline configuration), then time K more steps as extra.settled_window (the operating "
                   "point of the package settles ~2 s into the load); 0: skip")
    p.add_argument("--graph-train", type=int, default=1, help="1 (with --graph 1): the training pass is replayed from hipGraphs too")
    p.add_argument("--ddp-path", type=int, default=None, help="1: gradients through the flat buffers of train_step.FlatGradSync (two-call "
                   "backward, all-reduces between the graph segments) - the path every N > 1 rank takes; on one GPU the collectives "
                   "are no-ops, so N = 1 and N > 1 run the same code.  Default: 1 when --gpus > 1, else 0 (one backward call, one graph)")
    p.add_argument("--grad-sync", default=None, choices=("flat", "ddp"), help="N > 1: 'flat' (default) or torch DDP (eager training pass)")
    p.add_argument("--ab-steps", type=int, default=10, help="steps per leg of the interleaved kernel-set A/B behind the main measurement "
                   "(extra.ab: [default, round4, default, round4] in this process, on this box; ops.KERNEL_SETS); 0: skip.  Single-GPU "
                   "headline runs only")
    p.add_argument("--ab-order", default="default,round4,default,round4", help="the legs of the A/B (names of ops.KERNEL_SETS; each leg "
                   "starts from 'default' and applies its set on top)")
    p.add_argument("--strict", action="store_true", help="exit non-zero if any informational measurement (other_configs, model_kernel_"
                   "rooflines) failed; the failures are always visible as {'error': ...} entries and in extra.errors")
    return p.parse_args()


def cpu_baseline(args):
    """CPU oracle (numpy APGD restatement + plain-torch ConvNeXt-T-CvSt), full AT step, fp32."""
    import numpy as np
    from oracle import apgd_oracle as O
    from oracle import models_ref as M
    torch.manual_seed(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, args.cpu_threads))      # oversubscribing a 256-thread host is ~100x slower
    torch.set_num_threads(cores)
    model = M.build(args.arch, not_original=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    B = args.cpu_batch
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, args.res, args.res, generator=g)
    y = torch.randint(0, 1000, (B,), generator=g)

    def one_step():
        model.eval()
        xb, _, _, _, _ = O.apgd_train_oracle(O.TorchModelAdapter(model, y.numpy()), x.numpy(), y.numpy(), "Linf",
                                             args.eps, args.n_iter)
        model.train()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(torch.from_numpy(xb)), y)
        loss.backward()
        opt.step()

    times = []
    t_all = time.perf_counter()
    while len(times) < args.cpu_steps + 1 and (time.perf_counter() - t_all) < 40.0:
        t0 = time.perf_counter()
        one_step()
        times.append(time.perf_counter() - t0)
    timed = times[1:] if len(times) > 1 else times       # first step = warm-up unless it is the only one
    dt = sum(timed)
    return {"value": round(B * len(timed) / dt, 3), "unit": "img/s", "cores": cores,
            "kind": "port",
            "sample": f"{len(timed)} full AT step(s) (APGD-{args.n_iter} + train fwd/bwd + AdamW), batch {B}, "
                      f"{args.res}x{args.res}, fp32, oracle/apgd_oracle.py + oracle/models_ref.py, "
                      f"{cores} of {avail} host threads, {dt:.1f} s"}


def model_kernel_rooflines(R, dev, B, iters=10):
    """Live HIP-event timing (after the timed region) of the two hand-written model kernels that lead the step profile, on
    synthetic stage-0 tensors of this batch, against their algorithmic bytes / flops:
    fused LN+MLP forward at C = 96 and the rolling-window depthwise 7x7 forward at 56x56x96.  Back-to-back launches of one
    kernel run slower than the same kernel inside the step (profiles/: 271 / 148 us there), so avg_us is conservative; min_us is
    the fastest single launch."""
    import torch
    lib = R._lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    out, best = [], []

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        best.append(min(ts))
        return sum(ts) / len(ts)

    g = torch.Generator(device=dev).manual_seed(0)
    for C, HW in ((96, 56), (192, 28), (384, 14)):          # stages 0, 1 and 2 of ConvNeXt-T (2 / 2 / 9 of its blocks' attack passes)
        M = B * HW * HW
        u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
        x = torch.randn(M, C, device=dev, generator=g)
        w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
        w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
        lw, lb, b1, b2, gm = (torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(4 * C, device=dev),
                              torch.zeros(C, device=dev), torch.ones(C, device=dev))
        wf = R.ops._pack_mlp(w1, w2)
        o = torch.empty(M, C, device=dev)
        t = timed(lambda: R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, None, None, wf.data_ptr(),
                                                              b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), 0, o.data_ptr(), 0,
                                                              None, M, C, stream), "cnx_block_mlp_fwd"))
        nbytes, flops = M * C * (2 + 4 + 4), 16.0 * M * C * C
        out.append({"kernel": "fused LN+MLP forward C=%d (cnx_block_mlp_fwd, M=%d)" % (C, M), "avg_us": round(t * 1e6, 1),
                    "min_us": round(best[-1] * 1e6, 1),
                    "hbm": {"achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4),
                            "algorithmic_bytes_per_launch": nbytes},
                    "mfma": {"achieved": round(flops / t / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / t / 2.5e15, 4)}})
        if lib.cnx_block_mlp_hpre_supported(C):             # the attack's forward / input-gradient pair at this width
            wb = R.ops._pack_mlp_bwd(w1, w2)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            hp = torch.empty(lib.cnx_block_mlp_hpre_elems(M, C), device=dev, dtype=torch.bfloat16)
            gout = torch.randn(M, C, device=dev, generator=g)
            du = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
            tf = timed(lambda: R._lib.check(lib.cnx_block_mlp_fwd_hpre(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(),
                                                                        rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                                                                        gm.data_ptr(), x.data_ptr(), 0, o.data_ptr(), 0, hp.data_ptr(), M, C,
                                                                        stream), "cnx_block_mlp_fwd_hpre"))
            bf = best[-1]
            tb = timed(lambda: R._lib.check(lib.cnx_block_mlp_bwd_input_hpre(u.data_ptr(), lw.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                                              gout.data_ptr(), 0, gm.data_ptr(), wb.data_ptr(), hp.data_ptr(),
                                                                              du.data_ptr(), M, C, stream), "cnx_block_mlp_bwd_input_hpre"))
            for name, tt, bb, fl, nb in (("forward + Hpre workspace (cnx_block_mlp_fwd_hpre", tf, bf, flops, nbytes + 8 * M * C),
                                         ("input gradient from the workspace (cnx_block_mlp_bwd_input_hpre", tb, best[-1], flops,
                                          M * C * (4 + 2 + 2 + 8))):
                out.append({"kernel": "fused LN+MLP %s, C=%d, M=%d)" % (name, C, M), "avg_us": round(tt * 1e6, 1), "min_us": round(bb * 1e6, 1),
                            "hbm": {"achieved": round(nb / tt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nb / tt / 8e12, 4),
                                    "algorithmic_bytes_per_launch": nb},
                            "mfma": {"achieved": round(fl / tt / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / tt / 2.5e15, 4)}})
    C, HW = 96, 56
    M = B * HW * HW
    x = torch.randn(M, C, device=dev, generator=g)
    b2 = torch.zeros(C, device=dev)
    w49 = torch.randn(49, C, device=dev, generator=g) * 0.1
    xo = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: R._lib.check(lib.cnx_dwconv7x7_nhwc(x.data_ptr(), 0, w49.data_ptr(), b2.data_ptr(), None, xo.data_ptr(), 1, B, HW,
                                                           HW, C, 0, stream), "cnx_dwconv7x7_nhwc"))
    nbytes = M * C * (4 + 2)
    out.append({"kernel": "dwconv7x7_dma_kernel<float, bf16> (cnx_dwconv7x7_nhwc, %dx%dx%dx%d)" % (B, HW, HW, C),
                "avg_us": round(t * 1e6, 1), "min_us": round(best[-1] * 1e6, 1),
                "hbm": {"achieved": round(nbytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / t / 8e12, 4),
                        "algorithmic_bytes_per_launch": nbytes}})
    return out


def other_configs(R, dev, graph=1, which=("cfg3", "cfg4", "cfg5")):
    """Single-GPU, per-GPU-batch numbers of the other BASELINE.json configurations (informational, after the timed region):
    #3 ViT-B-CvSt APGD-2 AT @224 (per-GPU batch 256), #4 ConvNeXt-L-CvSt APGD-3 AT @320 (per-GPU batch 128),
    #5 the 100-step APGD-CE + APGD-T evaluation (`run_standard_evaluation`) on ConvNeXt-B-CvSt @224 (batch 100, fp32 as AA_eval.py runs it)."""
    import torch
    out = {}

    def at_step(arch, res, batch, n_iter, steps=3, warm=4):
        torch.manual_seed(0)
        model = R.get_new_model(arch, pretrained=False, not_original=True, img_size=res)
        tr = R.ATTrainStep(model, arch, R.AdvConfig(attack="apgd", norm="Linf", eps=4 / 255, n_iter=n_iter, graph=graph), dev, lr=1e-3,
                           amp_dtype=torch.bfloat16, ema=True)
        g = torch.Generator(device=dev).manual_seed(7)
        x = torch.rand(batch, 3, res, res, device=dev, generator=g)
        y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
        for _ in range(warm):
            tr.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # what the timed steps computed is checked, not only timed: the last step's loss is finite, and the attack of one more
        # (untimed) call hands the training pass a batch inside the eps-ball around x and inside [0, 1] (autopgd_train_clean.py:224-226)
        eps = 4 / 255
        z = tr._perturbed(x, y)
        dev_max = float((z.float() - x).abs().max())
        lo, hi = float(z.min()), float(z.max())
        res_d = {"img_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "per_gpu_batch": batch, "res": res,
                 "n_iter": n_iter, "steps": steps, "loss": round(float(loss), 4), "linf_dist": round(dev_max, 6),
                 "x_best_range": [round(lo, 6), round(hi, 6)]}
        bad = []
        if not torch.isfinite(loss).item():
            bad.append("non-finite loss")
        if not dev_max <= eps * (1 + 1e-5) + 1e-7:
            bad.append(f"|x_best - x|_inf = {dev_max:.6f} > eps = {eps:.6f}")
        if lo < 0.0 or hi > 1.0:
            bad.append(f"x_best leaves [0, 1]: [{lo}, {hi}]")
        if bad:
            res_d["error"] = "; ".join(bad)
        del tr, model, z
        torch.cuda.empty_cache()
        return res_d

    def note(msg):
        print(f"[bench other_configs +{time.perf_counter() - t_start:.0f}s] {msg}", file=sys.stderr, flush=True)
    t_start = time.perf_counter()
    for key, cfg in (("cfg3_vit_b_cvst_apgd2_at_224", ("vit_b", 224, 256, 2)),
                     ("cfg4_convnext_large_cvst_apgd3_at_320", ("convnext_large", 320, 128, 3))):
        if key[:4] not in which:
            continue
        try:
            out[key] = at_step(*cfg)
        except Exception as e:                               # informational: the headline line still prints; --strict fails the run
            out[key] = {"error": repr(e)}
        R.graphed.reset()                                    # the captured programs' graph pools go with the configuration
        torch.cuda.empty_cache()
        note(f"{key}: {out[key]}")
    if "cfg5" not in which:
        return out
    try:
        torch.manual_seed(0)
        model = R.get_new_model("convnext_base", pretrained=False, not_original=True).to(dev).to(memory_format=torch.channels_last).eval()
        g = torch.Generator(device=dev).manual_seed(9)
        bs = 100                                             # AA_eval.py's batch for the large models (BASELINE.json config #5)
        x = torch.rand(bs, 3, 224, 224, device=dev, generator=g)
        with torch.no_grad():
            y = model(x).argmax(1)                           # every point starts robust: APGD-CE runs on all of them
        # warm-up at the SAME batch size: MIOpen's find mode times every solver (its naive kernels included, 4 - 10 ms each)
        # the first time it meets a convolution shape - ~10 s for this model's fp32 problems, which a warm-up at
        # another batch size leaves inside the timed run (round 2's first cfg5 figure, 2.4 img/s, was mostly that)
        R.aa_eval.apgd_attack(model, x, y, "Linf", 4 / 255, 2, "ce", None, True, g)
        # ... and at the padded sizes the still-robust subsets are run at (run_standard_evaluation(buckets=True)): in an evaluation of
        # 50 batches every bucket is met in the first batches; here each is met once, untimed, with both losses
        yt = (y + 1) % 1000
        nb = bs
        while True:
            nb2 = R.aa_eval._bucket(max(1, (nb + 1) // 2), bs)
            if nb2 >= nb:
                break
            nb = nb2
            R.aa_eval.apgd_attack(model, x[:nb].contiguous(), y[:nb], "Linf", 4 / 255, 2, "ce", None, True, g)
            R.aa_eval.apgd_attack(model, x[:nb].contiguous(), y[:nb], "Linf", 4 / 255, 2, "dlr-targeted", yt[:nb], True, g)
        R.aa_eval.apgd_attack(model, x, y, "Linf", 4 / 255, 2, "dlr-targeted", yt, True, g)
        torch.cuda.synchronize()
        note("cfg5 warm-up done")
        t0 = time.perf_counter()
        # the config as AA_eval.py:226-239 drives it: APGD-CE, then APGD-T x 9 target classes on the points still robust.
        # (A random-init model is broken by APGD-CE almost everywhere, so the targeted runs see few points, each with a batch
        # size the libraries have not met: `attack_runs` / `sample_iters` say what the evaluation amounted to.)
        _, st = R.run_standard_evaluation(model, x, y, bs=bs, norm="Linf", eps=4 / 255, attacks_to_run=("apgd-ce", "apgd-t"),
                                          n_iter=100, n_target_classes=9, verbose=bool(os.environ.get("APGD_BENCH_VERBOSE")),
                                          buckets=True, graph=bool(graph))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["cfg5_convnext_base_cvst_standard_eval_ce_t_100step_224"] = {
            "img_s": round(bs / dt, 2), "s_per_batch": round(dt, 3), "batch": bs, "n_iter": 100, "dtype": "f32",
            "attack_runs": st["attack_runs"], "sample_iters": st["sample_iters"],
            "sample_iters_per_s": round(st["sample_iters"] / dt, 1), "robust_after": st["robust"],
            # a random-init model has no point left after APGD-CE at eps = 4/255: the line then times the CE leg alone
            "mode": "buckets + graph replay: a throughput mode whose real rows agree with the unpadded eager evaluation to rounding, not bit for "
                    "bit (aa_eval.run_standard_evaluation); robust accuracy is reported from buckets=False, graph=False",
            "legs": "CE + T" if st["attack_runs"] >= 2 else "CE only (no point survived APGD-CE: the targeted runs had nothing to attack)"}
        # the same evaluation at an eps small enough that about half the points survive APGD-CE, so that the targeted leg
        # (dlr-targeted on the product model, targets from the clean logits, still-robust subset) is inside a timed figure too
        try:
            eps_t = cfg5_eps_with_survivors(R, model, x, y, g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, st2 = R.run_standard_evaluation(model, x, y, bs=bs, norm="Linf", eps=eps_t, attacks_to_run=("apgd-ce", "apgd-t"),
                                               n_iter=100, n_target_classes=2, buckets=True, graph=bool(graph))
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            out["cfg5_convnext_base_cvst_standard_eval_ce_t_100step_224"]["with_targeted_leg"] = {
                "eps": eps_t, "n_target_classes": 2, "attack_runs": st2["attack_runs"], "sample_iters": st2["sample_iters"],
                "sample_iters_per_s": round(st2["sample_iters"] / dt2, 1), "robust_after": st2["robust"], "s_per_batch": round(dt2, 3)}
        except Exception as e:
            out["cfg5_convnext_base_cvst_standard_eval_ce_t_100step_224"]["with_targeted_leg"] = {"error": repr(e)}
        del model
        torch.cuda.empty_cache()
    except Exception as e:
        out["cfg5_convnext_base_cvst_standard_eval_ce_t_100step_224"] = {"error": repr(e)}
    note(f"cfg5: {out['cfg5_convnext_base_cvst_standard_eval_ce_t_100step_224']}")
    return out


def cfg5_eps_with_survivors(R, model, x, y, gen, n_iter=20):
    """An eps at which roughly half of the batch survives APGD-CE: bisection over short (20-iteration) runs, downwards from 4/255."""
    lo, hi, eps = 0.0, 4 / 255, 4 / 255
    for _ in range(7):
        eps = 0.5 * (lo + hi)
        _, acc, _, _ = R.aa_eval.apgd_attack(model, x, y, "Linf", eps, n_iter, "ce", None, True, gen)
        frac = float(acc.float().mean())
        if 0.3 <= frac <= 0.7:
            break
        if frac > 0.7:
            lo = eps
        else:
            hi = eps
    return eps


def ab_leg(R, args, dev, x, y, n_warm, order=("default", "round4", "default", "round4")):
    """The round's kernel work against the round-4 kernel set, in THIS process on THIS box (the boxes of the pool sit at different
    operating points - 1200 W / 2.2 GHz or 1010 W / 2.39 GHz - and differ by more than a round's kernel work): for each entry of
    `order` select the kernel set (ops.kernel_set - module switches + cnx_runtime_switch), drop every captured graph, build a fresh
    seeded model + ATTrainStep, run the same warm-up as the main measurement (captures included) and time `--ab-steps` steps with the
    same bracket.  Reports the per-leg times, power and clock, the medians, and shader cycles per step (ms x MHz), which takes the
    box's clock out of the comparison."""
    legs = []
    start = R.ops.kernel_set("default")
    try:
        for name in order:
            R.ops.kernel_set("default")
            R.ops.kernel_set(name)
            R.graphed.reset()
            torch.cuda.empty_cache()
            torch.manual_seed(0)
            model = R.get_new_model(args.arch, pretrained=False, not_original=True)
            tr = R.ATTrainStep(model, args.arch, R.AdvConfig(attack="apgd", norm="Linf", eps=args.eps, n_iter=args.n_iter, graph=args.graph),
                               dev, lr=1e-3, channels_last=True, amp_dtype=torch.bfloat16, ema=True,
                               mixup=object() if args.soft_labels else None, soft_targets=args.soft_labels, gemm_table=True,
                               graph_train=bool(args.graph) and bool(args.graph_train))
            for _ in range(n_warm):
                tr.step(x, y)
            torch.cuda.synchronize()
            power = PowerSampler(dev.index or 0)
            power.start()
            t0 = time.perf_counter()
            for _ in range(args.ab_steps):
                loss = tr.step(x, y)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.ab_steps * 1e3
            pw = power.stop() or {}
            captured = sum(v is not None for v in tr._tg.values())
            leg = {"set": name, "ms_per_step": round(ms, 3), "avg_W": pw.get("avg_W"), "avg_sclk_MHz": pw.get("avg_sclk_MHz"),
                   "mcycles_per_step": round(ms * pw["avg_sclk_MHz"] * 1e-3, 2) if pw.get("avg_sclk_MHz") else None,
                   "loss": round(float(loss), 4), "train_graph_captured": captured}
            if not torch.isfinite(loss).item():
                leg["error"] = "non-finite loss"
            legs.append(leg)
            del tr, model
    finally:
        R.ops.kernel_set(start)
        R.graphed.reset()
        torch.cuda.empty_cache()

    def med(key, name):
        v = sorted(l[key] for l in legs if l["set"] == name and l.get(key) is not None)
        return v[len(v) // 2] if len(v) % 2 else (round(0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3) if v else None)
    out = {"order": list(order), "steps_per_leg": args.ab_steps, "warmup_per_leg": n_warm, "legs": legs,
           "default_ms": med("ms_per_step", "default"), "r4_ms": med("ms_per_step", "round4"),
           "default_mcycles": med("mcycles_per_step", "default"), "r4_mcycles": med("mcycles_per_step", "round4"),
           "median_ms": {n: med("ms_per_step", n) for n in dict.fromkeys(order)},
           "median_mcycles": {n: med("mcycles_per_step", n) for n in dict.fromkeys(order)},
           "sclk_MHz": med("avg_sclk_MHz", "default"), "W": med("avg_W", "default"),
           "round4_set": "ops.KERNEL_SETS['round4']: library weight gradients (GEMM + ConvStem filter gradients; bias gradients by ops.conv_bias_grad), recomputing training backward, per-channel "
                         "gradient passes, separate tracking pass, single-wavefront C = 256 / 384 forward, head pool on NCHW, round-4 "
                         "depthwise strips - the end-of-round-4 kernel selection inside today's library"}
    if out["default_ms"] and out["r4_ms"]:
        out["r4_over_default"] = round(out["r4_ms"] / out["default_ms"], 4)
    return out


def spawn_ranks(args, child_argv=None, ndev=None) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: start N fresh rank processes (one per GPU, RCCL over
    xGMI; the reference's ``mp.spawn`` of ``main.py:1128-1152``), relay rank 0's JSON line, and fail if the job that ran is
    not an N-rank job.  The parent never touches the GPU (``torch.cuda.device_count()`` does not initialise HIP on this
    image) and never re-execs itself: the ranks are plain child processes.  ``child_argv`` / ``ndev`` let the CPU tests drive
    the same launcher with a gloo worker (tests/test_bench_launcher.py)."""
    n = args.gpus
    ndev = torch.cuda.device_count() if ndev is None else ndev
    if ndev < n:
        print(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    threads = os.environ.get("OMP_NUM_THREADS") or str(max(1, min(8, (os.cpu_count() or n) // n)))
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=threads,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # ranks > 0: stderr into a temporary file, shown if the job fails (a rank that dies at start-up used to be invisible)
        errs.append(None if r == 0 else tempfile.TemporaryFile())
        procs.append(subprocess.Popen(child_argv or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r]))
    # rank 0's pipe is drained by a thread while ALL children are polled under one deadline: when any rank exits non-zero (bad
    # device, RCCL failure, import error) the others - parked in init_process_group or a collective - are ended by handle and
    # the job fails at once instead of hanging on rank 0's pipe
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("APGD_BENCH_TIMEOUT_S", "1800"))
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes) or time.monotonic() > deadline:
            failed = True
            break
        time.sleep(0.05)
    if failed:
        for i, p in enumerate(procs):
            if codes[i] is None:
                p.kill()                                     # the exact children we started, by handle
                p.wait()
                codes[i] = -9
    reader.join(timeout=10)
    out0 = (chunks[0] if chunks else b"").decode()
    if any(codes):
        for r, f in enumerate(errs):
            if f is not None and codes[r] not in (0, -9):
                f.seek(0)
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n" + f.read().decode(errors="replace")[-4000:] + "\n")
    if any(codes):
        sys.stdout.write(out0)
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
            except ValueError:
                pass
    if line is None or line.get("n_gpus") != n:
        sys.stdout.write(out0)
        print(f"bench.py: expected one JSON line with n_gpus == {n}, got {None if line is None else line.get('n_gpus')}",
              file=sys.stderr)
        return 1
    print(json.dumps(line), flush=True)
    return 0


def mark_graph_fallbacks(extra, graph, graph_train):
    """``{"error": ...}`` entries for a step that ran on an eager fallback although replay was asked for: a failed (or missing)
    capture of the attack (``extra["attack_graph"]`` = graphed.STATS) or of the training pass (``extra["train_graph"]``).  Such an
    entry lands in ``extra.errors`` and makes ``--strict`` exit non-zero: under a process group a failed three-segment capture must
    not pass for a measurement of the replayed N > 1 step."""
    tg, ag = extra.get("train_graph") or {}, extra.get("attack_graph") or {}
    if graph_train and (tg.get("failed") or not tg.get("captured")):
        tg["error"] = ("the training pass was NOT replayed from hipGraphs (capture failed or never happened): this line measured "
                       "the eager training pass")
    if graph and (ag.get("failed") or not ag.get("captures")):
        ag["error"] = "the attack was NOT replayed from hipGraphs (capture failed or never happened): this line measured the eager attack"
    return extra


class PowerSampler:
    """Package power and shader clock of this rank's GPU from sysfs hwmon (power1_input in uW, freq1_input in Hz), sampled by a
    host thread during the timed steps.  The fused LN+MLP kernels run AT the package power cap (profiles/r02_power_and_overlap.md),
    where a kernel's duration is its energy, so the step's average power is part of reading the numbers.  None if unreadable."""

    def __init__(self, dev_index):
        self.dir, self.samples, self._stop, self._th = None, [], False, None
        try:
            import glob
            p = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(d).endswith(bdf):
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
        except Exception:                                       # noqa: BLE001 - telemetry is optional
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except Exception:                                       # noqa: BLE001
            return None

    def _run(self):
        while not self._stop:
            w, hz = self._read("power1_input"), self._read("freq1_input")
            if w is not None:
                self.samples.append((w * 1e-6, (hz or 0.0) * 1e-6))
            time.sleep(0.02)

    def start(self):
        if self.dir:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join()
        if not self.samples:
            return None
        ws, fs = [a for a, _ in self.samples], [b for _, b in self.samples]
        cap = self._read("power1_cap")
        return {"avg_W": round(sum(ws) / len(ws), 1), "max_W": round(max(ws), 1), "cap_W": round(cap * 1e-6, 1) if cap else None,
                "avg_sclk_MHz": round(sum(fs) / len(fs), 1), "samples": len(ws)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import revisiting_at_amd as R
    from revisiting_at_amd import apgd as apgd_mod
    import torch.distributed as dist

    rank, local, world = R.setup_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (spawns "
                         f"the ranks itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU fallback for the product path"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    R._lib.load()
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True                      # main.py:25

    model = R.get_new_model(args.arch, pretrained=False, not_original=True)
    adv = R.AdvConfig(attack="apgd", norm="Linf", eps=args.eps, n_iter=args.n_iter, graph=args.graph)
    ddp_path = (world > 1) if args.ddp_path is None else bool(args.ddp_path)
    grad_sync = args.grad_sync or ("flat" if ddp_path else None)
    if world > 1 and grad_sync is None:
        raise SystemExit("bench.py: N > 1 ranks need a gradient exchange (--ddp-path 1 or --grad-sync ddp)")
    trainer = R.ATTrainStep(model, args.arch, adv, dev, lr=1e-3, distributed=world > 1, channels_last=True,
                            amp_dtype=torch.bfloat16, ema=True, mixup=object() if args.soft_labels else None,
                            soft_targets=args.soft_labels, gemm_table=True,
                            graph_train=bool(args.graph) and bool(args.graph_train), grad_sync=grad_sync)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    B = args.batch
    x = torch.rand(B, 3, args.res, args.res, device=dev, generator=g)        # synthetic 224x224x3 batch in [0,1)
    if args.soft_labels:
        y = torch.softmax(torch.randn(B, 1000, device=dev, generator=g), dim=1)
    else:
        y = torch.randint(0, 1000, (B,), device=dev, generator=g)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the attack's hipGraphs are captured on its third call (graphed.WARMUP_CALLS eager calls first): never inside the timed region
    # ... and the training pass on the step after TRAIN_GRAPH_WARMUP eager ones
    n_warm = max(args.warmup, R.graphed.WARMUP_CALLS + 1, R.train_step.TRAIN_GRAPH_WARMUP + 1) if args.graph else args.warmup
    if world > 1:
        # Library find / autotune (MIOpen's per-shape search, hipBLASLt's heuristics) is per process: eight ranks would each spend
        # the same seconds on it, at once, on one host.  Rank 0 meets every shape first - one local training forward / backward,
        # no optimizer step, no collective - and leaves its results in the user find-db; the others follow behind a barrier.
        t_lib = time.perf_counter()
        if rank == 0:
            trainer.warm_libraries(x, y)
            torch.cuda.synchronize()
        dist.barrier()
        if rank != 0:
            trainer.warm_libraries(x, y)
            torch.cuda.synchronize()
        dist.barrier()
        t_lib = time.perf_counter() - t_lib
    for _ in range(n_warm):
        trainer.step(x, y)
    sync()
    reduces0 = trainer.sync.reduces if trainer.sync else 0
    apgd_mod.PROFILE_EVENTS = []                                # K1 launches get bracketed by HIP events
    power = PowerSampler(dev.index or 0) if rank == 0 else None
    if power:
        power.start()
    t0 = time.perf_counter()
    t_call = []
    for _ in range(args.steps):
        tc = time.perf_counter()
        trainer.step(x, y)
        t_call.append(time.perf_counter() - tc)
    dt_enqueue = time.perf_counter() - t0                      # host time inside the K step() calls (no synchronisation in a step)
    sync()
    dt = time.perf_counter() - t0
    reduces_per_step = ((trainer.sync.reduces - reduces0) / args.steps) if trainer.sync else None
    power_stats = power.stop() if power else None
    events = apgd_mod.PROFILE_EVENTS
    apgd_mod.PROFILE_EVENTS = None
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # roofline of the dominant hand-written kernel (APGD Linf update), its two forms reported separately:
    #   general (i >= 1): 20 algorithmic B/elem - the `roofline` object;
    #   first   (i == 0): x_adv_old aliases x_adv, 16 algorithmic B/elem - `roofline.first_iter`.
    # Since round 5 the launch also performs the row moves of the iteration before it (K3; at i = 0 the prologue's clones): its
    # algorithmic bytes are the step's 20 (16) B/element plus 4 B/element per row-move destination the launch writes (from the flag
    # bytes each timed launch read; the moves' sources are the step's own operands) - at fp32, so with int8 gradient signs the kernel
    # moves FEWER bytes than that and `frac` can pass 1 while `frac_moved`, the physical reading, is ~0.74; `frac_sum_8d` adds 8d's K3
    # rule literally; `frac_step_only` keeps the step's 20 (16) B/element over the same time for comparison with rounds 1-4.
    # `achieved` = algorithmic bytes / HIP-event time measured here; `traffic` = HBM bytes per launch of the same kernel
    # form from the PMC passes (profiles/k1_traffic.json).  With int8 gradient signs the kernel moves 17 / 13 B/elem, i.e.
    # LESS than the algorithmic figure: `bytes_moved` and `moved_GBs` state that side by side.
    n_elem = B * 3 * args.res * args.res
    traffic_tab = {}
    try:
        traffic_tab = json.load(open(os.path.join(ROOT, "profiles", "k1_traffic.json"))).get("variants", {})
    except Exception:
        pass

    E_row = 3 * args.res * args.res

    def track_bytes(flags, gbytes, first):
        """Bytes of the row moves a fused update launch performs, from the flag bytes the launch read: (algorithmic, algorithmic by
        the literal sum of SURVEY.md 8d's K1 and K3 rules, moved, per-flag sample counts).
        algorithmic = what the FUSED operation has to move at fp32: the row moves' sources are the step's own operands (read once),
        so a flagged sample adds 4 B/elem per destination written (x_best, grad_best, x_best_adv; a restore: x_adv and grad = 8).
        8d's K3 rule prices the tracking as a pass of its own (8 B/elem read per flagged sample + 4 per destination; a restore
        8 + 8): its sum with K1's 20 B/elem counts the shared reads twice - reported next to it as `frac_sum_8d`.
        Iteration 0 writes the prologue's three clones (4 B/elem each; their source is the step's own operand)."""
        if first:
            return 12.0 * n_elem, 12.0 * n_elem, (8.0 + gbytes) * n_elem, (B, B, 0)
        if flags is None:
            return 0.0, 0.0, 0.0, (0, 0, 0)
        f = flags.to(torch.int64)
        nb, mc, hv = (f & 1) != 0, (f & 2) != 0, (f & 4) != 0
        rs = hv & ~nb
        anyf = nb | mc                                        # (a restore-only sample: its 8 + 8 are counted below, no more)
        n_nb, n_mc, n_rs, n_any = int(nb.sum()), int(mc.sum()), int(rs.sum()), int(anyf.sum())
        alg = (8.0 * n_nb + 4.0 * n_mc + 8.0 * n_rs) * E_row
        alg8d = (8.0 * n_any + 8.0 * n_nb + 4.0 * n_mc + 16.0 * n_rs) * E_row
        # moved: the step reads x_adv / grad anyway; NEW_BEST writes 4 + g, MISCLS 4, a restore reads 4 + g and writes 4
        mov = ((4.0 + gbytes) * n_nb + 4.0 * n_mc + (8.0 + gbytes) * n_rs) * E_row
        return alg, alg8d, mov, (n_nb, n_mc, n_rs)

    def k1_entry(sel, alg_bpe, form):
        ev = [(a.elapsed_time(b), gb, name, fl) for (name, i, a, b, gb, fl) in events if name.startswith("apgd_linf_step") and sel(i)]
        if not ev:
            return None
        avg_ms = sum(e[0] for e in ev) / len(ev)
        gbytes = ev[0][1]
        fused = ev[0][2] == "apgd_linf_step_track_f32"
        blk = gbytes == 1 and R.ops.SIGN_BLOCKED and E_row % 1024 == 0 and not fused          # the stem's blocked sign order
        step_alg, step_mov = alg_bpe * n_elem, (alg_bpe - 4 + gbytes) * n_elem
        tb = [track_bytes(e[3], gbytes, form == "first") if fused else (0.0, 0.0, 0.0, (0, 0, 0)) for e in ev]
        alg_launch = step_alg + sum(t[0] for t in tb) / len(tb)                              # average per launch
        alg8d_launch = step_alg + sum(t[1] for t in tb) / len(tb)
        mov_launch = step_mov + sum(t[2] for t in tb) / len(tb)
        ach = alg_launch / (avg_ms * 1e-3) / 1e9
        gname = ('i8blk' if blk else 'i8') if gbytes == 1 else 'f32'
        traffic, tr, traffic_src = None, {}, None
        for key in ([f"track_{form}_{gname}_f3", f"track_{form}_{gname}_f1", f"track_{form}_{gname}"] if fused else [f"{form}_{gname}"]):
            tr = traffic_tab.get(key, {})
            if tr and abs(alg8d_launch - tr.get("algorithmic_bytes_per_launch", -1)) < 1.0:
                traffic, traffic_src = tr.get("hbm_bytes_per_launch"), f"PMC pass of this kernel form ({key})"
                break
        if traffic is None and fused and form == "general" and n_elem == 256 * 3 * 224 * 224:
            # a launch's flag bytes are a MIX (most samples NEW_BEST | MISCLS, some NEW_BEST only, ...): the PMC passes cover the
            # uniform mixes (every sample 3, every sample 1) and the plain step; the kernel's traffic is linear in the per-flag sample
            # counts (PMC / designed bytes = 1.0000 in every pass), so the launch's traffic is assembled from the passes' per-sample costs
            t0, t1, t3 = (traffic_tab.get(k, {}).get("hbm_bytes_per_launch") for k in
                          (f"general_{gname}", f"track_general_{gname}_f1", f"track_general_{gname}_f3"))
            if t0 and t1 and t3:
                nbr, mcr, rsr = (sum(t[3][k] for t in tb) / len(tb) for k in range(3))
                traffic = t0 + nbr * (t1 - t0) / B + mcr * (t3 - t1) / B + rsr * (8.0 + gbytes) * E_row
                traffic_src = ("PMC passes of the plain step and of the uniform flag mixes 1 / 3, combined by this run's average per-flag "
                               f"sample counts per launch (NEW_BEST {nbr:.1f}, MISCLS {mcr:.1f}, restore {rsr:.1f} of {B})")
        if traffic is None and rank == 0:
            print(f"bench.py: roofline.traffic is null for the {form} K1 form - profiles/k1_traffic.json holds PMC passes of the headline "
                  f"shape and of uniform flag bytes only (this run: {alg_launch:.0f} algorithmic bytes per launch); "
                  "re-run tools/k1_traffic.py for this shape", file=sys.stderr)
        if fused:
            kern = f"linf_step_track_vec4_kernel<{'true' if form == 'first' else 'false'}> (apgd_linf_step_track_f32: the Linf step + "
            kern += ("the prologue's clones" if form == "first" else "the row moves of the iteration before it")
            kern += f", {'int8 sign' if gbytes == 1 else 'fp32'} gradient)"
        else:
            kern = ((("linf_step_i8blk_kernel<false>" if form == "general" else "linf_step_i8blk_kernel<true>") if blk else
                     ("linf_step_vec4_kernel" if form == "general" else "linf_step_first_vec4_kernel"))
                    + f" (apgd_linf_step_f32, {('blocked int8 sign' if blk else 'int8 sign') if gbytes == 1 else 'fp32'} gradient)")
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic and round(traffic), "traffic_source": traffic_src, "kernel": kern,
                "launches": len(ev), "avg_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": round(alg_launch),
                # the step's 20 (16) B/element + 4 B/element per row-move destination written, from the flag bytes each
                # launch read; `frac_sum_8d`: with 8d's K3 rule added literally (it re-counts the sources the step already reads)
                "algorithmic_bytes_step": step_alg, "algorithmic_bytes_row_moves": round(alg_launch - step_alg),
                "frac_sum_8d": round(alg8d_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # the algorithmic bytes AT THE DTYPES THE LAUNCH RUNS WITH (int8 gradient signs: 17 (13) B/element for the step, + 4 / 4 / 1
                # B/element per destination written, a restore 8 + 1): what the bus has to carry, so this reading cannot pass 1.0 while
                # `frac` (SURVEY 8d's fp32 accounting) can
                "algorithmic_bytes_as_run": round(mov_launch), "frac_as_run": round(mov_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_moved": round(mov_launch), "moved_GBs": round(mov_launch / (avg_ms * 1e-3) / 1e9, 1),
                # the PHYSICAL reading: bytes the kernel is designed to move (= the PMC traffic) / time / 8 TB/s
                "frac_moved": round(mov_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # the step alone at its 20 (16) B/element over the same time (what rounds 1-4 reported for the unfused kernel)
                "frac_step_only": round(step_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    roof = k1_entry(lambda i: i > 0, K1_BYTES_PER_ELEM, "general")
    first = k1_entry(lambda i: i == 0, K1_BYTES_PER_ELEM_IT0, "first")
    if roof is None:
        roof, first = first, None
    if roof is not None and first is not None:
        roof["first_iter"] = first

    # Steady state.  For the first ~2 s under this load the package runs at another operating point than from then on (1140 - 1200 W
    # at 2.19 - 2.24 GHz shader clock, then ~1030 W at 2.38 GHz: K = 20 steps behind 5 / 15 / 40 warm-up steps take 54.4 / 53.5 /
    # 51.7 ms per step), so the contract's window - K steps right behind W warm-up steps, the `value` of this
    # line - sits inside that transient.  `extra.settled_window` is the same measurement (same bracket, same K1 events) after the
    # device has run `--settle-steps` steps in all: what a training run sees after its first two seconds.  Same counts on every rank.
    settled = None
    done = n_warm + args.steps
    if args.settle_steps > 0:
        for _ in range(max(0, args.settle_steps - done)):
            trainer.step(x, y)
        sync()
        events_first = events
        apgd_mod.PROFILE_EVENTS = []
        power2 = PowerSampler(dev.index or 0) if rank == 0 else None
        if power2:
            power2.start()
        t0s = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(x, y)
        sync()
        ts = torch.tensor([time.perf_counter() - t0s], device=dev, dtype=torch.float64)
        p2 = power2.stop() if power2 else None
        events = apgd_mod.PROFILE_EVENTS
        apgd_mod.PROFILE_EVENTS = None
        if world > 1:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        k1s = k1_entry(lambda i: i > 0, K1_BYTES_PER_ELEM, "general")
        settled = {"img_s": round(world * B * args.steps / float(ts.item()), 2), "ms_per_step": round(float(ts.item()) / args.steps * 1e3, 3),
                   "steps": args.steps, "behind_steps": max(done, args.settle_steps),
                   "k1_avg_us": k1s and k1s["avg_us"], "k1_frac": k1s and k1s["frac"], "k1_frac_moved": k1s and k1s["frac_moved"],
                   "package_power": p2}
        events = events_first

    # Host cost of ENQUEUEING a step = the median step() call among the first steps of the timed region, when the device queue
    # is still shallow.  The whole-loop average (host_loop_ms_per_step) also contains back-pressure: once the host is ~7 steps
    # (~9000 kernels) ahead, the runtime blocks launches until the device drains, and the loop then runs at the DEVICE's pace
    # whatever the host costs (20-step runs: 35 ms per step in the loop, 12 ms per unblocked call).
    head = sorted(t_call[:min(len(t_call), 6)])
    extra = {"settled_window": settled, "host_enqueue_ms_per_step": round(head[len(head) // 2] * 1e3, 3),
             "host_loop_ms_per_step": round(dt_enqueue / args.steps * 1e3, 3), "package_power": power_stats,
             "attack_graph": dict(R.graphed.STATS, enabled=bool(args.graph)), "gemm_mode": R.ops._GEMM_MODE,
             # what the host issues per steady-state step: graph launches (attack segments + the training pass), the K1 kernel
             # launches between the attack's segments, input copies; the device executes ~1250 kernels per step from them
             # (profiles/r03_step.md).  Eager steps issue ~900 kernel launches from Python instead.
             "host_launches_per_step": ({"graph_launches": sum(pr.n_graphs for en in R.graphed._programs.values()
                                                                 for pr in [en.get("prog")] if pr is not None)
                                         + sum(v.n_graphs for v in trainer._tg.values() if v is not None),
                                         "kernel_launches": sum(len(pr.steps) - pr.n_graphs for en in R.graphed._programs.values()
                                                                for pr in [en.get("prog")] if pr is not None),
                                         "collectives": (sum(len(v.steps) - v.n_graphs for v in trainer._tg.values() if v is not None)
                                                         if world > 1 else 0),
                                         "copies": 3} if args.graph else None),
             "train_graph": {"enabled": bool(trainer.graph_train),
                             "captured": sum(v is not None for v in trainer._tg.values()), "failed": sum(v is None for v in trainer._tg.values()),
                             # graph segments of the captured training pass: 1 on one GPU, 3 with the two all-reduces between them
                             "segments": [v.n_graphs for v in trainer._tg.values() if v is not None]},
             # how the ranks' gradients meet: "flat" = train_step.FlatGradSync (same replayed step at every N), "ddp" = torch DDP
             "grad_sync": {"path": grad_sync, "allreduce_bytes_per_step": trainer.sync.bytes_per_step if trainer.sync else None,
                           "groups_MB": [round(4e-6 * b.numel(), 1) for b in trainer.sync.flat] if trainer.sync else None,
                           "reduces_per_step": reduces_per_step, "cut": bool(trainer.sync and trainer.sync.cut is not None),
                           "library_warmup_s": round(t_lib, 2) if world > 1 else None}}
    # A capture that failed leaves the step on its eager fallback (correct, but 30 - 40 ms of host per step): that must not pass for a
    # measurement of the replayed step - an {"error": ...} entry puts it into extra.errors, and --strict fails the run.
    mark_graph_fallbacks(extra, bool(args.graph), bool(trainer.graph_train))
    if True:
        # attack-only throughput (same tensors, eval mode), a few repetitions
        base = trainer.inner.base_model
        base.eval()
        reps = max(2, min(args.steps, 5))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            R.apgd_train(base, x, y, norm="Linf", eps=args.eps, n_iter=args.n_iter,
                         mixup=object() if args.soft_labels else None)
            sync()
            ta = time.perf_counter()
            for _ in range(reps):
                R.apgd_train(base, x, y, norm="Linf", eps=args.eps, n_iter=args.n_iter,
                             mixup=object() if args.soft_labels else None)
            sync()
            extra["attack_only_img_s"] = round(world * B * reps / (time.perf_counter() - ta), 1)
        base.train()
        if rank == 0 and world == 1 and args.arch.startswith("convnext") and R.ops.MODE != "eager":
            try:
                extra["model_kernel_rooflines"] = model_kernel_rooflines(R, dev, B)
            except Exception as e:                       # informational: the line still prints; --strict fails the run
                extra["model_kernel_rooflines"] = {"error": repr(e)}
        if power_stats and power_stats.get("avg_sclk_MHz"):
            # shader cycles per step of the contract's window: the box's operating point taken out of ms_per_step
            extra["mcycles_per_step"] = round(dt / args.steps * 1e3 * power_stats["avg_sclk_MHz"] * 1e-3, 2)
        if rank == 0 and world == 1 and args.ab_steps > 0 and R.ops.MODE != "eager" and args.arch.startswith("convnext") and not args.attack_only:
            trainer = model = base = None                    # the headline model, its optimizer state and graphs go first
            R.graphed.reset()
            torch.cuda.empty_cache()
            try:
                extra["ab"] = ab_leg(R, args, dev, x, y, n_warm, tuple(args.ab_order.split(",")))
            except Exception as e:                           # informational: the line still prints; --strict fails the run
                extra["ab"] = {"error": repr(e)}
            print(f"[bench ab] {extra['ab']}", file=sys.stderr, flush=True)
        if (rank == 0 and world == 1 and args.arch == "convnext_tiny" and R.ops.MODE != "eager" and not args.no_other_configs
                and (args.other_configs is not None or not args.no_cpu_baseline)):
            del trainer, model, base, x, y
            R.graphed.reset()                            # the headline model's captured attack (its graph pool) goes first
            torch.cuda.empty_cache()
            extra["other_configs"] = other_configs(R, dev, args.graph, tuple((args.other_configs or "cfg3,cfg4,cfg5").split(",")))
        extra["ops_mode"] = R.ops.MODE
        extra["device"] = torch.cuda.get_device_name(dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    def errors_in(o, path=""):
        if isinstance(o, dict):
            return ([path] if "error" in o else []) + [e for k, v in o.items() for e in errors_in(v, f"{path}/{k}")]
        if isinstance(o, list):
            return [e for i, v in enumerate(o) for e in errors_in(v, f"{path}[{i}]")]
        return []
    extra["errors"] = errors_in(extra)

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "adversarial images/sec, ConvNeXt-T-CvSt APGD-2 AT @224" if args.arch == "convnext_tiny"
                      and args.n_iter == 2 and args.res == 224 else
                      f"adversarial images/sec, {args.arch}-CvSt APGD-{args.n_iter} AT @{args.res}",
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.arch}-CvSt APGD-{args.n_iter} adversarial-training step, eps={args.eps:.6f} Linf, "
                                   f"{args.res}x{args.res}x3 fp32 inputs, bf16 autocast, per-GPU batch {B}, "
                                   f"{'soft (mixup-style)' if args.soft_labels else 'hard'} labels, AdamW + EMA, "
                                   f"{('batch sharded over ' + str(world) + ' ranks, gradient all-reduce over RCCL') if world > 1 else 'single GPU'}",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if args.strict and extra["errors"]:
        print(f"bench.py --strict: failed measurements: {extra['errors']}", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
