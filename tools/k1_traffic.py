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
    for (name, _), v in per.items():
        out.setdefault(name, []).append(v)
    return out


def pick(tab, *subs):
    for name, vals in tab.items():
        if all(s in name for s in subs):
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
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
