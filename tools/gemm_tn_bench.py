#!/usr/bin/env python3
"""cnx_gemm_tn (D = A^T B over the rows of two row-major bf16 operands) against the library composition it replaces
(ops._wgrad: split-K batched hipBLASLt GEMM + partial-sum pass) on the weight-gradient shapes of ConvNeXt-T / ViT-B at batch 256;
HIP events around 10 back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
from revisiting_at_amd import ops
lib = R._lib.load()
S = torch.cuda.current_stream().cuda_stream


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(802816, 384, 96), (802816, 96, 384), (200704, 768, 192), (200704, 192, 768), (50176, 1536, 384), (50176, 384, 1536),
          (12544, 3072, 768), (12544, 768, 3072), (50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for M, N1, N2 in shapes:
    A = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    B = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    D = torch.empty(N1, N2, device="cuda")
    ws = torch.empty(max(4, lib.cnx_gemm_tn_ws_floats(M, N1, N2)), device="cuda")
    own = t(lambda: lib.cnx_gemm_tn(A.data_ptr(), N1, B.data_ptr(), N2, D.data_ptr(), ws.data_ptr(), M, N1, N2, S))
    ref = t(lambda: ops._wgrad_lib(A, B))
    err = float((D - ops._wgrad_lib(A, B)).norm() / D.norm())
    gf = 2.0 * M * N1 * N2 / 1e9
    mb = M * (N1 + N2) * 2 / 1e6
    print(f"M={M:7d} N1={N1:5d} N2={N2:5d}: cnx_gemm_tn {own:7.1f} us ({gf / own * 1e3:6.0f} TFLOP/s, operands {mb / own:5.2f} TB/s, "
          f"splits x partial {ws.numel() * 4 / 1e6:6.1f} MB) | library bmm + sum {ref:7.1f} us | rel diff {err:.1e}", flush=True)

# the tile-layout forms of the training pass (cnx_gemm_tn_ex: one operand in CNX_TN_ACC tiles, column sums of A): dW2 = dO^T H and
# dW1 = dHpre^T LN(u) at the three fused stages
print("cnx_gemm_tn_ex, one operand in accumulator-order tiles (timing only: the tile operand is random bits of small bf16 values)")
for M, C in ((802816, 96), (200704, 192), (50176, 384)):
    rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    tiles = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
    for name, a, lda, la, b, ldb, lb, N1, N2 in (("dW2 = dO^T H  ", rows, C, 0, tiles, 0, 1, C, 4 * C), ("dW1 = dHpre^T a", tiles, 0, 1, rows, C, 0, 4 * C, C)):
        D = torch.empty(N1, N2, device="cuda"); cs = torch.empty(N1, device="cuda")
        ws = torch.empty(max(4, lib.cnx_gemm_tn_ws_floats(M, N1, N2)), device="cuda")
        us = t(lambda: lib.cnx_gemm_tn_ex(a.data_ptr(), lda, la, b.data_ptr(), ldb, lb, D.data_ptr(), cs.data_ptr(), ws.data_ptr(), M, N1, N2, S))
        print(f"M={M:7d} C={C:4d} {name}: {us:7.1f} us (operands {M * 5 * C * 2 / 1e6 / us:5.2f} TB/s)", flush=True)
