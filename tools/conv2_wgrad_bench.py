#!/usr/bin/env python3
"""Filter gradient of the second ConvStem convolution (48 -> 96, 3x3 / 2) at batch 256 on the 112 x 112 map: cnx_conv3x3s2_wgrad against the
library's convolution_backward on the same operands; HIP events, median of 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CI, CO, H, W = 48, 96, 112, 112
x = torch.randn(B, H, W, CI, device="cuda").to(torch.bfloat16)
dy = torch.randn(B, H // 2, W // 2, CO, device="cuda").to(torch.bfloat16)
w = (torch.randn(CO, CI, 3, 3, device="cuda") * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
dw = torch.empty(CO, 3, 3, CI, device="cuda"); db = torch.empty(CO, device="cuda")
ws = torch.empty(lib.cnx_conv3x3s2_wgrad_ws_floats(CI, CO), device="cuda")
S = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3


t_own = timeit(lambda: lib.cnx_conv3x3s2_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, W, CI, CO, S))
t_lib = timeit(lambda: torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, [CO], [2, 2], [1, 1], [1, 1], False,
                                                           [0, 0], 1, [False, True, True]))
gf = 2.0 * B * (H // 2) * (W // 2) * 9 * CI * CO / 1e9
mb = (x.numel() + dy.numel()) * 2 / 1e6
print(f"B={B}: cnx_conv3x3s2_wgrad {t_own:.1f} us ({gf / t_own * 1e3:.0f} TFLOP/s, {mb / t_own:.2f} TB/s of operands) | library {t_lib:.1f} us")
