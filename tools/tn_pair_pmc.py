#!/usr/bin/env python3
"""A few launches of cnx_gemm_tn_pair (and of the two single cnx_gemm_tn_ex launches) at one block shape, for rocprofv3 --pmc passes.
usage: tools/tn_pair_pmc.py [C=384] [M=50176] [reps=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import revisiting_at_amd as R
lib = R._lib.load()
S = torch.cuda.current_stream().cuda_stream
C = int(sys.argv[1]) if len(sys.argv) > 1 else 384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50176
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
N1, N2 = 4 * C, C
a_rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
do_rows = torch.randn(M, C, device="cuda").to(torch.bfloat16)
dhp = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
h = (torch.randn(M * 4 * C, device="cuda") * 0.1).to(torch.bfloat16)
dw1 = torch.empty(N1, N2, device="cuda"); db1 = torch.empty(N1, device="cuda")
dw2 = torch.empty(N2, N1, device="cuda"); db2 = torch.empty(N2, device="cuda")
ws1 = torch.empty(max(lib.cnx_gemm_tn_ws_floats(M, N1, N2), lib.cnx_gemm_tn_ws_floats(M, N2, N1)), device="cuda")
wsp = torch.empty(lib.cnx_gemm_tn_pair_ws_floats(M, N1, N2), device="cuda")
for _ in range(reps):
    R._lib.check(lib.cnx_gemm_tn_pair(dhp.data_ptr(), a_rows.data_ptr(), C, h.data_ptr(), do_rows.data_ptr(), C, dw1.data_ptr(), db1.data_ptr(),
                                      dw2.data_ptr(), db2.data_ptr(), wsp.data_ptr(), M, N1, N2, S), "pair")
    lib.cnx_gemm_tn_ex(do_rows.data_ptr(), C, 0, h.data_ptr(), 0, 1, dw2.data_ptr(), db2.data_ptr(), ws1.data_ptr(), M, N2, N1, S)
    lib.cnx_gemm_tn_ex(dhp.data_ptr(), 0, 1, a_rows.data_ptr(), C, 0, dw1.data_ptr(), db1.data_ptr(), ws1.data_ptr(), M, N1, N2, S)
torch.cuda.synchronize()
print("operand bytes per pair launch", M * 10 * C * 2, "partials", wsp.numel() * 4)
