#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 (rocpd sqlite) result: durations from `kernels`, counters from `counters_collection`.
Usage: python tools/pmc_dump.py path/to/results.db [name-filter]"""
import collections
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = con.cursor()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)[:70]


rows = cur.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels").fetchall()
agg = collections.OrderedDict()
for n, d, g, w, lds, v, a, s, sc in rows:
    if flt and flt not in n:
        continue
    k = (short(n), g, w, lds, v, a, s, sc)
    agg.setdefault(k, []).append(d)
for k, ds in agg.items():
    ds.sort()
    print(f"{k[0]:70s} grid {k[1]:>9} wg {k[2]:>4} lds {k[3]:>6} vgpr {k[4]:>3} agpr {k[5]:>3} sgpr {k[6]:>3} scratch {k[7]}  n={len(ds)} med {ds[len(ds)//2]/1e3:.1f} us")
try:
    rows = cur.execute("select kernel_name, grid_size, counter_name, value, dispatch_id from counters_collection").fetchall()
except Exception:
    rows = []
cc = collections.OrderedDict()
for n, g, c, v, d in rows:
    if flt and flt not in n:
        continue
    cc.setdefault((short(n), g), collections.OrderedDict()).setdefault(c, {}).setdefault(d, 0.0)
    cc[(short(n), g)][c][d] += v
for k, cs in cc.items():
    print(f"\n{k[0]} grid {k[1]}")
    for c, per in cs.items():
        vals = sorted(per.values())
        print(f"   {c:28s} median/dispatch {vals[len(vals)//2]:.4g}")
