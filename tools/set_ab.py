#!/usr/bin/env python3
"""Interleaved A/B of two kernel sets on the benchmarked AT step in one process (the same legs bench.py's `extra.ab` runs):
usage: tools/set_ab.py SET_B [pairs=3] [steps=15] [SET_A=default]"""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
a = sys.argv[4] if len(sys.argv) > 4 else "default"
order = ",".join([a, b] * pairs)
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs",
                      "--ab-order", order, "--ab-steps", str(steps)], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
ab = d["extra"]["ab"]
print("main", d["ms_per_step"], "ms", d["extra"].get("mcycles_per_step"), "Mc")
for leg in ab["legs"]:
    print({k: leg[k] for k in leg if k in ("set", "ms_per_step", "avg_W", "avg_sclk_MHz", "mcycles_per_step", "loss")})
print({k: v for k, v in ab.items() if k != "legs"})
