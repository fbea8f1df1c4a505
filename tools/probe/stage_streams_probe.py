#!/usr/bin/env python3
"""ConvNeXt-T stage 2 (9 blocks, C = 384, 256 x 14 x 14) attack forward + input gradient: one batch vs batch chunks on streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import revisiting_at_amd as R

C, N, H = 384, 256, 14
torch.manual_seed(0)
stage = R.architecture.ConvNeXtStage(C, C, 9, True, 1e-6).cuda().eval()
x = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
cot = torch.randn_like(x)


_side = {}


def run_batch_chunks(blocks, x, k):
    """blocks over k batch chunks of x, chunk i on stream i (current + k-1 side streams), issued block by block, joined on the
    current stream.  (Weight packs are already cached by the one-stream warm-up: a product version needs stream-aware caches.)"""
    main = torch.cuda.current_stream()
    if k not in _side:
        _side[k] = [torch.cuda.Stream() for _ in range(k - 1)]
    streams = [main] + _side[k]
    xs = list(x.chunk(k, 0))
    ev = main.record_event()
    for s in streams[1:]:
        s.wait_event(ev)
        x.record_stream(s)
    for blk in blocks:
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                xs[i] = blk(xs[i])
    for xi, s in zip(xs[1:], streams[1:]):
        main.wait_stream(s)
        xi.record_stream(main)
    return torch.cat(xs, 0)


def once(k):
    xd = x.detach().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16), R.ops.attack_forward():
        y = run_batch_chunks(stage.blocks, xd, k) if k > 1 else stage.blocks(xd)
    with R.ops.input_grad_only():
        torch.autograd.grad(y, xd, cot)


for k in (1, 2, 3, 4, 1, 3):
    for _ in range(3):
        once(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(10):
        once(k)
    th = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    print(f"chunks={k}: GPU {e0.elapsed_time(e1) / 10:.3f} ms per fwd+bwd of the stage, host enqueue {th * 100:.3f} ms", flush=True)
