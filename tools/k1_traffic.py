#!/usr/bin/env python3
"""profiles/k1_traffic.json from the two rocprofv3 PMC passes over tools/k1_pmc.py:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o k1 -- python3 tools/k1_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o k1 -- python3 tools/k1_pmc.py
    python3 tools/k1_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/k1_traffic.json

Counters are in KiB per dispatch; FETCH_SIZE is corrected by the factor measured on the calibration copy in the same pass
(0.5 on gfx950 for wide coalesced reads -> x2), WRITE_SIZE by its own (1.0)."""
import csv
import glob
import json
import os
import statistics
import sys

csv.field_size_limit(1 << 30)
B, E = 256, 3 * 224 * 224
TENSOR_KIB = B * E * 4 / 1024.0


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter:
            continue
        per.setdefault((row["Kernel_Name"], int(row["Dispatch_Id"])), 0.0)
        per[(row["Kernel_Name"], int(row["Dispatch_Id"]))] += float(row["Counter_Value"])
    out = {}
    for (name, _), v in sorted(per.items(), key=lambda kv: kv[0][1]):      # per kernel name, in dispatch order
        out.setdefault(name, []).append(v)
    return out


def pick(tab, *subs, part=None):
    """median over the dispatches of the kernel whose name holds every ``subs``; ``part`` = (k, n): the k-th of n equal slices
    of its dispatches in launch order (tools/k1_pmc.py launches the fused kernel's flag mixes one after the other)"""
    for name, vals in tab.items():
        if all(s in name for s in subs):
            if part is not None:
                k, n = part
                m = len(vals) // n
                vals = vals[k * m:(k + 1) * m]
            return statistics.median(vals)
    raise KeyError(subs)


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
f_cal = pick(fetch, "copyBuffer") / TENSOR_KIB if any("copyBuffer" in k for k in fetch) else pick(fetch, "direct_copy") / TENSOR_KIB
w_cal = pick(write, "copyBuffer") / TENSOR_KIB if any("copyBuffer" in k for k in write) else pick(write, "direct_copy") / TENSOR_KIB
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/k1_pmc.py; B=256, E=150528",
       "fetch_calibration": f_cal, "write_calibration": w_cal, "variants": {}}
forms = {"general_f32": ("linf_step_vec4_kernel<float", 20), "general_i8": ("linf_step_vec4_kernel<signed char", 17),
         "first_f32": ("linf_step_first_vec4_kernel<float", 16), "first_i8": ("linf_step_first_vec4_kernel<signed char", 13),
         "general_i8blk": ("linf_step_i8blk_kernel<false>", 17), "first_i8blk": ("linf_step_i8blk_kernel<true>", 13)}
for key, (sub, bpe) in forms.items():
    try:
        fk, wk = pick(fetch, sub), pick(write, sub)
    except KeyError:
        continue
    res["variants"][key] = {"fetch_kib_raw": fk, "write_kib_raw": wk,
                            "hbm_bytes_per_launch": (fk / f_cal + wk / w_cal) * 1024.0,
                            "bytes_moved_by_design": bpe * B * E,
                            "algorithmic_bytes_per_launch": (16 if key.startswith("first") else 20) * B * E}
# the fused update + row-move kernel (round 5).  algorithmic bytes: the step's 20 (16) B/element + SURVEY 8d's K3 rule (a flagged
# sample: 8 read + 4 per destination written; iteration 0: the prologue's three clones); by design it moves 17 (13) + the writes
N = B * E
fused = {"track_general_f32_f3": ("linf_step_track_vec4_kernel<float, false>", (0, 2), 20 + 8 + 12, 20 + 12),
         "track_general_f32_f1": ("linf_step_track_vec4_kernel<float, false>", (1, 2), 20 + 8 + 8, 20 + 8),
         "track_first_f32": ("linf_step_track_vec4_kernel<float, true>", None, 16 + 12, 16 + 12),
         "track_general_i8_f3": ("linf_step_track_vec4_kernel<signed char, false>", (0, 2), 20 + 8 + 12, 17 + 9),
         "track_general_i8_f1": ("linf_step_track_vec4_kernel<signed char, false>", (1, 2), 20 + 8 + 8, 17 + 5),
         "track_first_i8": ("linf_step_track_vec4_kernel<signed char, true>", None, 16 + 12, 13 + 9)}
for key, (sub, part, alg, mov) in fused.items():
    try:
        fk, wk = pick(fetch, sub, part=part), pick(write, sub, part=part)
    except KeyError:
        continue
    res["variants"][key] = {"fetch_kib_raw": fk, "write_kib_raw": wk, "hbm_bytes_per_launch": (fk / f_cal + wk / w_cal) * 1024.0,
                            "bytes_moved_by_design": mov * N, "algorithmic_bytes_per_launch": alg * N}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
