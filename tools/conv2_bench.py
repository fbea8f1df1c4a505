#!/usr/bin/env python3
"""Second ConvStem convolution (3x3, stride 2, 48 -> 96 / 64 -> 96) through the C ABI vs the library convolution: results and time.
Usage: python tools/conv2_bench.py [CI] [batch] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import revisiting_at_amd as R
lib = R._lib.load()
CI = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 112
CO, W = 96, H
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H, W, CI, device="cuda", generator=g).to(torch.bfloat16)
w = torch.randn(CO, CI, 3, 3, device="cuda", generator=g) * (CI * 9) ** -0.5
b = torch.randn(CO, device="cuda", generator=g) * 0.1
pk = torch.empty(lib.cnx_conv3x3s2_packed_elems(CI, CO), device="cuda", dtype=torch.bfloat16)
R._lib.check(lib.cnx_conv3x3s2_pack(w.data_ptr(), 0, pk.data_ptr(), CI, CO, S), "pack")
out = torch.empty(B, H // 2, W // 2, CO, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(B, H // 2, W // 2, CO, device="cuda", generator=g).to(torch.bfloat16)
dx = torch.empty(B, H, W, CI, device="cuda", dtype=torch.bfloat16)


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[0]


fwd = lambda: R._lib.check(lib.cnx_conv3x3s2_fwd(x.data_ptr(), pk.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, CI, CO, S), "fwd")
bwd = lambda: R._lib.check(lib.cnx_conv3x3s2_dgrad(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), B, H, W, CI, CO, S), "dgrad")
fwd(); bwd(); torch.cuda.synchronize()
xn = x.permute(0, 3, 1, 2).float().requires_grad_()            # NCHW view of the NHWC tensor, fp32 math on the bf16 values
ref = F.conv2d(xn, w.to(torch.bfloat16).float(), b, stride=2, padding=1)
(gref,) = torch.autograd.grad(ref, xn, dy.permute(0, 3, 1, 2).float())
e_f = float((out.permute(0, 3, 1, 2).float() - ref).norm() / ref.norm())
e_b = float((dx.permute(0, 3, 1, 2).float() - gref).norm() / gref.norm())
xl = x.permute(0, 3, 1, 2)                                       # channels-last bf16, as the model hands it to the library
wl = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
libf = lambda: F.conv2d(xl, wl, b.to(torch.bfloat16), stride=2, padding=1)
yl = libf()
dyl = dy.permute(0, 3, 1, 2)
libb = lambda: torch.ops.aten.convolution_backward(dyl, xl, wl, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
tf, tb, lf, lb_ = timeit(fwd), timeit(bwd), timeit(libf), timeit(libb)
nbytes = x.numel() * 2 + out.numel() * 2
print(f"CI={CI} B={B} {H}x{W}: fwd {tf[0]:.1f} / {tf[1]:.1f} us ({nbytes / tf[0] / 1e6:.2f} TB/s) rel err {e_f:.2e} | library {lf[0]:.1f} us || "
      f"dgrad {tb[0]:.1f} / {tb[1]:.1f} us rel err {e_b:.2e} | library {lb_[0]:.1f} us")
