#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the depthwise 7x7 kernels: forward fp32 -> bf16 and input gradient bf16 -> fp32 + add at the
ConvNeXt-T stage shapes, batch 256 (APGD_DW_WIN=0 / 1 selects the LDS-ring / sliding-window kernels), plus a device copy of known
size as the byte-count calibration of FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
S = torch.cuda.current_stream().cuda_stream
B = 256
shapes = ((96, 56), (192, 28), (384, 14))
if len(sys.argv) > 1:
    shapes = tuple(tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(","))
for C, HW in shapes:
    w = torch.randn(49, C, device="cuda"); b = torch.randn(C, device="cuda")
    x32 = torch.randn(B, HW, HW, C, device="cuda"); xb = torch.randn(B, HW, HW, C, device="cuda").bfloat16()
    add = torch.randn(B, HW, HW, C, device="cuda")
    ob = torch.empty(B, HW, HW, C, device="cuda", dtype=torch.bfloat16); o32 = torch.empty(B, HW, HW, C, device="cuda")
    for _ in range(8):
        assert lib.cnx_dwconv7x7_nhwc(x32.data_ptr(), 0, w.data_ptr(), b.data_ptr(), None, ob.data_ptr(), 1, B, HW, HW, C, 0, S) == 0
        o32.copy_(x32)                                    # calibration: n * 4 bytes read, n * 4 written
        assert lib.cnx_dwconv7x7_nhwc(xb.data_ptr(), 1, w.data_ptr(), None, add.data_ptr(), o32.data_ptr(), 0, B, HW, HW, C, 1, S) == 0
torch.cuda.synchronize()
print("done")
