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


PAIR_ONLY = "--pair-only" in sys.argv
if PAIR_ONLY:
    sys.argv.remove("--pair-only")
shapes = [(802816, 384, 96), (802816, 96, 384), (200704, 768, 192), (200704, 192, 768), (50176, 1536, 384), (50176, 384, 1536),
          (12544, 3072, 768), (12544, 768, 3072), (50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for M, N1, N2 in ([] if PAIR_ONLY else shapes):
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
for M, C in (() if PAIR_ONLY else ((802816, 96), (200704, 192), (50176, 384))):
    rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    tiles = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
    for name, a, lda, la, b, ldb, lb, N1, N2 in (("dW2 = dO^T H  ", rows, C, 0, tiles, 0, 1, C, 4 * C), ("dW1 = dHpre^T a", tiles, 0, 1, rows, C, 0, 4 * C, C)):
        D = torch.empty(N1, N2, device="cuda"); cs = torch.empty(N1, device="cuda")
        ws = torch.empty(max(4, lib.cnx_gemm_tn_ws_floats(M, N1, N2)), device="cuda")
        us = t(lambda: lib.cnx_gemm_tn_ex(a.data_ptr(), lda, la, b.data_ptr(), ldb, lb, D.data_ptr(), cs.data_ptr(), ws.data_ptr(), M, N1, N2, S))
        print(f"M={M:7d} C={C:4d} {name}: {us:7.1f} us (operands {M * 5 * C * 2 / 1e6 / us:5.2f} TB/s)", flush=True)

# both weight gradients of a block: two cnx_gemm_tn_ex launches (+ two sums) against one cnx_gemm_tn_pair launch (+ one sum)
print("block weight gradients: 2 x cnx_gemm_tn_ex against cnx_gemm_tn_pair (alternating medians)")


def med(fn, it=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]


for M, C in ((802816, 96), (200704, 192), (50176, 384), (200704, 128), (50176, 256)):
    N1, N2 = 4 * C, C
    a_rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    do_rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    dhp = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
    h = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
    dw1 = torch.empty(N1, N2, device="cuda"); db1 = torch.empty(N1, device="cuda")
    dw2 = torch.empty(N2, N1, device="cuda"); db2 = torch.empty(N2, device="cuda")
    ws1 = torch.empty(max(lib.cnx_gemm_tn_ws_floats(M, N1, N2), lib.cnx_gemm_tn_ws_floats(M, N2, N1)), device="cuda")
    wsp = torch.empty(lib.cnx_gemm_tn_pair_ws_floats(M, N1, N2), device="cuda")

    def two():
        lib.cnx_gemm_tn_ex(do_rows.data_ptr(), C, 0, h.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws1.data_ptr(), M, N2, N1, S)
        lib.cnx_gemm_tn_ex(dhp.data_ptr(), 0, 1, a_rows.data_ptr(), C, 0, dw1.data_ptr(), db1.data_ptr(), ws1.data_ptr(), M, N1, N2, S)

    def pair():
        R._lib.check(lib.cnx_gemm_tn_pair(dhp.data_ptr(), a_rows.data_ptr(), C, h.data_ptr(), do_rows.data_ptr(), C, dw1.data_ptr(), db1.data_ptr(),
                                          dw2.data_ptr(), db2.data_ptr(), wsp.data_ptr(), M, N1, N2, S), "pair")
    two(); r = (dw1.clone(), dw2.clone())
    pair()
    err = max(float((dw1 - r[0]).norm() / r[0].norm()), float((dw2 - r[1]).norm() / r[1].norm()))
    res = []
    for _ in range(3):
        row = [med(two)]
        for mode in (0, 2, 3):
            lib.cnx_runtime_switch(5, mode)
            row.append(med(pair))
        res.append(row)
    lib.cnx_runtime_switch(5, 1)
    t2, tb, tr_, tpp = (sorted(v[i] for v in res)[1] for i in range(4))
    gf = 4.0 * M * N1 * N2 / 1e9
    print(f"M={M:7d} C={C:4d}: two launches {t2:7.1f} us ({gf / t2 * 1e3:5.0f} TF/s; partials {ws1.numel() * 4 / 1e6 * 2:6.1f} MB) | pair: two buffers {tb:7.1f} us "
          f"({gf / tb * 1e3:5.0f} TF/s) | ring {tr_:7.1f} us | two buffers, hand-over one k-step early {tpp:7.1f} us ({gf / tpp * 1e3:5.0f} TF/s) | "
          f"partials {wsp.numel() * 4 / 1e6:6.1f} MB | rel diff {err:.1e}", flush=True)
