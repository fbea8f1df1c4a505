"""Per-kernel register / scratch / LDS table of the HIP library, from hipcc's own remarks.

    python tools/resource_usage.py [file.hip ...] [--scratch-only] [--jobs N]

Compiles each source of ``revisiting-at_amd/csrc`` for gfx950 with ``-Rpass-analysis=kernel-resource-usage`` (device pass
only, with the Makefile's flags) and prints one row per kernel.  ``collect()`` is what ``tests/test_no_scratch.py`` calls: a
kernel of the default path with ScratchSize > 0 spills registers into memory inside its loop - round 2 shipped three of
those in the headline configuration without noticing.
"""
import argparse
import concurrent.futures as cf
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "revisiting-at_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "-Wno-unused-function", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]

_FIELD = re.compile(r"remark:\s+(?:Function )?([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis")


def sources():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^SRCS := (.*)$", mk, re.M)
    return [os.path.join(CSRC, f) for f in m.group(1).split()]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def analyse(path, extra=()):
    """-> list of dicts {file, name, sgpr, vgpr, agpr, scratch, occupancy, vgpr_spill, sgpr_spill, lds}"""
    r = subprocess.run([HIPCC] + FLAGS + list(extra) + [path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {path}:\n{r.stderr[-2000:]}")
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = _FIELD.search(line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Name":
            cur = {"file": os.path.basename(path), "name": val}
            rows.append(cur)
        elif cur is not None:
            k = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize": "scratch", "Occupancy": "occupancy",
                 "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size": "lds"}.get(key)
            if k:
                cur[k] = int(val)
    names = demangle([r_["name"] for r_ in rows])
    for r_ in rows:
        r_["name"] = names[r_["name"]].replace("(anonymous namespace)::", "")
    return rows


def collect(files=None, jobs=4):
    files = files or sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        out = []
        for rows in ex.map(analyse, files):
            out.extend(rows)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--scratch-only", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    rows = collect([os.path.abspath(f) for f in a.files] or None, a.jobs)
    print(f"{'file':22s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'scr':>4s} {'spill':>5s} {'occ':>3s} {'lds':>6s}  kernel")
    bad = 0
    for r in rows:
        if r.get("scratch", 0) > 0:
            bad += 1
        elif a.scratch_only:
            continue
        print(f"{r['file']:22s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} {r.get('scratch', 0):4d} "
              f"{r.get('vgpr_spill', 0):5d} {r.get('occupancy', 0):3d} {r.get('lds', 0):6d}  {r['name'][:150]}")
    print(f"{len(rows)} kernels, {bad} with scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
