#!/usr/bin/env python3
"""Per-kernel medians of rocprofv3 --pmc counters from the *_counter_collection.csv files under the given directories.
Usage: python tools/pmc_csv.py dir [dir ...] [--filter substring]"""
import csv
import glob
import os
import re
import statistics
import sys

csv.field_size_limit(1 << 30)
flt = ""
dirs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--filter":
        flt = args.pop(0)
    else:
        dirs.append(a)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:80]


tab = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if flt and flt not in row["Kernel_Name"]:
                continue
            key = (short(row["Kernel_Name"]), row["Counter_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for (k, c, _), v in per.items():
            tab.setdefault(k, {}).setdefault(c, []).append(v)
for k, cs in tab.items():
    print(k)
    for c, vals in sorted(cs.items()):
        print(f"    {c:32s} median {statistics.median(vals):.6g}   (n={len(vals)})")
