#!/usr/bin/env python3
"""Where the time of one ConvNeXt-T-CvSt pass goes, per stage and per operator (B=256, 224x224, bf16 autocast).

For every stage shape: depthwise-7x7+LN and the MLP (eager hipBLASLt composition and the fused MFMA kernel),
forward / input-gradient-only backward / full backward; then whole-model eval forward, forward + input
gradient (one attack iteration) and train forward + backward.  Usage: python tools/block_bench.py [--batch 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import revisiting_at_amd as R
from revisiting_at_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda")
B = args.batch


def timeit(fn, iters=args.iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def fwd_bwd(make_out, inputs, params, mode):
    """mode: 'fwd' (no_grad), 'in' (input grad only), 'full'."""
    if mode == "fwd":
        def run():
            with torch.no_grad():
                make_out()
        return run

    def run():
        for p in params:
            p.grad = None
        torch.clear_autocast_cache()          # casts cached under no_grad would cut inputs / weights out of the graph
        out = make_out()
        g = torch.ones_like(out)
        if mode == "in":
            with ops.input_grad_only():
                torch.autograd.grad(out, inputs, g)
        else:
            out.backward(g)
    return run


torch.manual_seed(0)
print(f"device {torch.cuda.get_device_name(0)}  batch {B}")
tot = {"fwd": 0.0, "in": 0.0, "full": 0.0}
for C, HW, depth in ((96, 56, 3), (192, 28, 3), (384, 14, 9), (768, 7, 3)):
    blk = R.architecture.ConvNeXtBlock(C).to(dev)
    x = torch.randn(B, C, HW, HW, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    params = list(blk.parameters())
    M = B * HW * HW
    flops = 16 * M * C * C
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xr = x.detach().permute(0, 2, 3, 1).contiguous().requires_grad_()
        u_bf = torch.randn(M, C, device=dev).to(torch.bfloat16)
        lib = R._lib.load()
        y_ln = ops.dwconv_ln(xr.detach(), blk.conv_dw.weight, blk.conv_dw.bias, blk.norm.weight, blk.norm.bias, 1e-6).detach().requires_grad_()

        def dw():
            return ops.dwconv_ln(xr, blk.conv_dw.weight, blk.conv_dw.bias, blk.norm.weight, blk.norm.bias, 1e-6)

        def mlp_eager():
            y = F.linear(F.gelu(F.linear(y_ln, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias)
            return xr.detach() + y * blk.gamma

        def tail_fused():
            wf = ops._cached((blk.mlp.fc1.weight, blk.mlp.fc2.weight), "mlp_packed", ops._pack_mlp)
            out = torch.empty(M, C, device=dev)
            R._lib.check(lib.cnx_block_mlp_fwd(u_bf.data_ptr(), blk.norm.weight.data_ptr(), blk.norm.bias.data_ptr(), 1e-6,
                                               None, None, wf.data_ptr(), blk.mlp.fc1.bias.data_ptr(),
                                               blk.mlp.fc2.bias.data_ptr(), blk.gamma.data_ptr(), xr.data_ptr(), 0,
                                               out.data_ptr(), 0, None, M, C, torch.cuda.current_stream().cuda_stream), "x")
            return out

        def block():
            return blk(x)

        row = [f"C={C:4d} HW={HW:2d} M={M:7d} x{depth}"]
        if ops.block_fused_supported(C):
            t = timeit(fwd_bwd(tail_fused, [], [], "fwd"))
            row.append(f"   LN+MLP fused kernel fwd {t * 1e3:8.1f} us  [{flops / t / 1e9:6.0f} TF/s]")
        for name, fn, inp in (("dwln", dw, [xr]), ("mlp_eager", mlp_eager, [y_ln]), ("block", block, [x])):
            t = {m: timeit(fwd_bwd(fn, inp, params, m)) for m in ("fwd", "in", "full")}
            extra = ""
            if name.startswith("mlp"):
                extra = f"  [{flops / t['fwd'] / 1e9:6.0f} TF/s fwd]"
            row.append(f"   {name:10s} fwd {t['fwd'] * 1e3:8.1f} us | fwd+in {t['in'] * 1e3:8.1f} us | fwd+full {t['full'] * 1e3:8.1f} us{extra}")
            if name == "block":
                for m in t:
                    tot[m] += depth * t[m]
        print("\n".join(row), flush=True)
print(f"sum over blocks x depth: fwd {tot['fwd']:.2f} ms | fwd+in {tot['in']:.2f} ms | fwd+full {tot['full']:.2f} ms")

model = R.get_new_model("convnext_tiny", pretrained=False, not_original=True).to(dev).to(memory_format=torch.channels_last)
xin = torch.rand(B, 3, 224, 224, device=dev).requires_grad_()
x = xin
params = list(model.parameters())
with torch.autocast("cuda", dtype=torch.bfloat16):
    model.eval()
    t_f = timeit(fwd_bwd(lambda: model(x), [x], params, "fwd"))
    t_i = timeit(fwd_bwd(lambda: model(x), [x], params, "in"))
    model.train()
    t_t = timeit(fwd_bwd(lambda: model(x.detach()), [x], params, "full"))
    stem = model.stem
    t_sf = timeit(fwd_bwd(lambda: stem(x), [x], list(stem.parameters()), "fwd"))
    t_si = timeit(fwd_bwd(lambda: stem(x), [x], list(stem.parameters()), "in"))
print(f"whole model: eval fwd {t_f:.2f} ms | fwd + input grad {t_i:.2f} ms | train fwd+bwd {t_t:.2f} ms")
print(f"stem: fwd {t_sf:.2f} ms | fwd + input grad {t_si:.2f} ms")
print(f"AT step estimate (K=2): {t_f + 2 * t_i + t_t:.2f} ms")
