#!/usr/bin/env python3
"""Per-workgroup phase timeline of the fused LN+MLP forward (needs a library built with -DMLP_ABLATE=1, which stamps the 100 MHz
wall clock at: 0 start, 1 LayerNorm done / hidden loop starts, 2 wave 0 leaves the hidden loop, 3 workgroup barrier passed,
4 stores issued; slot 7 = XCC_ID << 32 | HW_ID).  Answers: how many workgroups does a CU hold, how long is each phase under load,
and for what share of the kernel does a CU have NO workgroup inside the hidden loop (= the matrix pipe has nothing to do).
Usage: APGD_HIP_LIB=.../libapgd_abl.so python tools/blk_trace.py [--C 96] [--hw 56] [--batch 256]"""
import argparse
import ctypes as C_
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import revisiting_at_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=96)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--resid", default="f32")
args = ap.parse_args()
lib = R._lib.load()
dev = torch.device("cuda")
C, M = args.C, args.batch * args.hw * args.hw
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
x = torch.randn(M, C, device=dev, generator=g)
if args.resid == "bf16":
    x = x.to(torch.bfloat16)
w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
b1, b2, gm = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
wf = R.ops._pack_mlp(w1, w2)
out, mean, rstd = torch.empty(M, C, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
code = R._lib.dtype_code


def run():
    R._lib.check(lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(),
                                       b1.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), code(x.dtype), out.data_ptr(), 0, None, M, C, S),
                 "fwd")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
n_wg = min(8192, (M + 127) // 128)
SL = 12
buf = np.zeros(n_wg * SL, dtype=np.uint64)
lib.cnx_dbg_blk_trace.restype = C_.c_int
lib.cnx_dbg_blk_trace.argtypes = [C_.c_void_p, C_.c_int]
rc = lib.cnx_dbg_blk_trace(buf.ctypes.data, n_wg)
assert rc == 0, rc
t = buf.reshape(n_wg, SL)
hw = t[:, 7]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ts = t[:, :5].astype(np.int64)
t_begin = ts[:, 0].min()
ts = (ts - t_begin) * 0.01                      # microseconds (100 MHz)
span = ts[:, 4].max()
ph = {"prologue (load u, LayerNorm)": ts[:, 1] - ts[:, 0], "hidden loop": ts[:, 2] - ts[:, 1], "barrier after loop": ts[:, 3] - ts[:, 2],
      "epilogue (resid, store)": ts[:, 4] - ts[:, 3], "whole workgroup": ts[:, 4] - ts[:, 0]}
res = {"C": C, "M": M, "workgroups": int(n_wg), "event_us": round(e0.elapsed_time(e1) * 1e3, 1), "trace_span_us": round(float(span), 1),
       "distinct_cus": int(len(np.unique(cuid))), "phases_us": {}}
for k, v in ph.items():
    res["phases_us"][k] = {"p10": round(float(np.percentile(v, 10)), 2), "median": round(float(np.median(v)), 2),
                           "p90": round(float(np.percentile(v, 90)), 2)}
cyc = (t[:, 6].astype(np.int64) - t[:, 5].astype(np.int64)).astype(np.float64)
res["shader_clock_GHz_in_the_hidden_loop"] = round(float(np.median(cyc / np.maximum(ph["hidden loop"], 1e-3))) * 1e-3, 3)
res["hidden_loop_cycles_per_slice"] = round(float(np.median(cyc)) / (C // 8), 1)
if t[:, 10].any():                                  # pipelined loop: wave 0's cycles waiting for the weight DMA / in s_barrier / working
    res["hidden_loop_cycles_per_slice_split"] = {k: round(float(np.median(t[:, j].astype(np.float64))) / (C // 8 + 1), 1)
                                                 for k, j in (("vmcnt_wait", 8), ("barrier", 9), ("work", 10))}
# per-CU occupancy and coverage on a 0.1 us grid
grid = np.arange(0.0, span, 0.1)
resident, in_loop_any, in_loop_cnt = [], [], []
for c in np.unique(cuid):
    idx = np.nonzero(cuid == c)[0]
    occ = np.zeros_like(grid)
    loop = np.zeros_like(grid)
    for i in idx:
        occ += (grid >= ts[i, 0]) & (grid < ts[i, 4])
        loop += (grid >= ts[i, 1]) & (grid < ts[i, 2])
    live = occ > 0
    resident.append(occ[live].mean())
    in_loop_any.append((loop[live] > 0).mean())
    in_loop_cnt.append(loop[live].mean())
res["per_cu"] = {"resident_workgroups_mean": round(float(np.mean(resident)), 2), "share_of_time_with_a_workgroup_in_the_hidden_loop":
                 round(float(np.mean(in_loop_any)), 3), "workgroups_in_the_hidden_loop_mean": round(float(np.mean(in_loop_cnt)), 2),
                 "workgroups_per_cu": round(n_wg / len(np.unique(cuid)), 1)}
# the first CU's schedule, for the eye
c0 = np.unique(cuid)[0]
idx = np.nonzero(cuid == c0)[0]
idx = idx[np.argsort(ts[idx, 0])]
res["first_cu_schedule_us"] = [[round(float(v), 1) for v in ts[i]] for i in idx[:16]]
print(json.dumps(res))
