#!/usr/bin/env python3
"""Summarise ONE steady-state AT step from a rocprofv3 kernel trace (csv): kernels between the first
APGD-update launch of step `--step` and of the next step.  Usage:
  python tools/step_breakdown.py gpurun_out/prof2/bench_kernel_trace.csv [--step 3] [--top 40] [--md out.md]"""
import argparse
import collections
import csv
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--step", type=int, default=3)
ap.add_argument("--n-iter", type=int, default=2)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--md", default=None)
ap.add_argument("--title", default="")
ap.add_argument("--seq", default=None, help="also write the window's launches in order (start offset us, duration us, name)")
a = ap.parse_args()

rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [s for s, e, n in rows if "linf_step" in n]
t0, t1 = marks[a.step * a.n_iter], marks[(a.step + 1) * a.n_iter]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if t0 <= s < t1:
        agg[n][0] += 1
        agg[n][1] += e - s


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk_") or n.startswith("Custom_Cijk"):
        m = re.search(r"MT\d+x\d+x\d+", n)
        return "hipBLASLt " + n[:14] + "_" + (m.group(0) if m else "")
    if n.startswith("at::native::") or "at::native" in n[:60]:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
        k = re.findall(r"(\w+(?:Kernel|Functor|kernel_cuda|functor)\w*)", n)
        return "aten " + (m.group(1) if m else "") + (" " + k[1] if len(k) > 1 else "")
    if "ck::" in n or n.startswith("_ZN2ck"):
        m = re.search(r"kernel_\w+", n)
        return "CK " + (m.group(0)[:60] if m else n[:40])
    return n[:100]


grp = collections.defaultdict(lambda: [0, 0])
for n, (c, t) in agg.items():
    g = grp[short(n)]
    g[0] += c
    g[1] += t
tot = sum(t for _, t in grp.values())
GROUPS = (("fused LN+MLP (`blk_mlp_*`, `blk2_*`)", r"blk_mlp_|blk2_"), ("depthwise 7x7 (forward, input gradient, filter gradient)", r"dwconv7x7"),
          ("hipBLASLt / rocBLAS GEMMs", r"^hipBLASLt|Cijk_"), ("`cnx_gemm_nt`", r"gemm_nt_kernel"),
          ("`cnx_gemm_tn` weight gradients (+ split sums)", r"gemm_tn_"), ("MIOpen / CK convolutions", r"^CK |igemm_|naive_conv|miopen|Conv|wrw"),
          ("LayerNorm kernels", r"layernorm_"),
          ("element-wise tails around the library GEMMs + partial sums", r"gelu_|scale_residual|sum_parts|reduce_parts|colsum|block_dgamma|block_dln"),
          ("ConvStem convolutions (forward, input and filter gradients)", r"stem_conv|conv2_|wgrad_reduce"), ("attention", r"attn_"),
          ("attack kernels (K1, loss, state, tracking, init)", r"linf_step|track_rows|state_update|ce_pred|init_kernel|l2_step|dlr"),
          ("ATen copies / casts / optimizer / other", r""))
gt = collections.OrderedDict((g, [0, 0]) for g, _ in GROUPS)
for n, (c, t) in grp.items():
    for g, pat in GROUPS:
        if re.search(pat, n):
            gt[g][0] += c
            gt[g][1] += t
            break
lines = [f"step window {(t1 - t0) / 1e6:.2f} ms, GPU busy {tot / 1e6:.2f} ms, {sum(c for c, _ in grp.values())} launches", "",
         "| ms | % | launches | group |", "|---|---|---|---|"]
for g, (c, t) in sorted(gt.items(), key=lambda kv: -kv[1][1]):
    if c:
        lines.append(f"| {t / 1e6:.2f} | {100 * t / tot:.1f} | {c} | {g} |")
lines += ["", "| ms | % | calls | avg us | kernel |", "|---|---|---|---|---|"]
for n, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1])[: a.top]:
    lines.append(f"| {t / 1e6:.2f} | {100 * t / tot:.1f} | {c} | {t / c / 1e3:.1f} | `{n}` |")
# idle time of the device around every APGD-update launch of the window (the attack's graph segments end / begin there) and the
# largest gaps anywhere: how much of the window is launch latency rather than kernels
win = [(s_, e_, n_) for s_, e_, n_ in rows if t0 <= s_ < t1]
busy_until, gaps = t0, []
for s_, e_, n_ in win:
    if s_ > busy_until:
        gaps.append((s_ - busy_until, n_))
    busy_until = max(busy_until, e_)
lines += ["", f"idle inside the window: {sum(g for g, _ in gaps) / 1e3:.1f} us in {len(gaps)} gaps; largest: "
          + ", ".join(f"{g / 1e3:.1f} us before `{short(n_)[:40]}`" for g, n_ in sorted(gaps, reverse=True)[:6])]
out = "\n".join(lines)
print(out)
if a.seq:
    with open(a.seq, "w") as f:
        for s_, e_, n_ in win:
            f.write(f"{(s_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:7.1f} {short(n_)[:110]}\n")
if a.md:
    with open(a.md, "w") as f:
        f.write(f"# {a.title}\n\n{out}\n")
