#!/usr/bin/env python3
"""Run one kernel back to back for a few seconds and poll rocm-smi (power, clocks) meanwhile: is the kernel power-capped?
Usage: python tools/probe/power_watch.py [--C 96] [--hw 56] [--seconds 6] [--what fwd|k1|gemm]"""
import argparse, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import revisiting_at_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=96)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--what", default="fwd")
args = ap.parse_args()
lib = R._lib.load()
dev = torch.device("cuda")
C, M = args.C, 256 * args.hw * args.hw
S = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
u = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
x = torch.randn(M, C, device=dev, generator=g)
w1 = torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5
w2 = torch.randn(C, 4 * C, device=dev, generator=g) * (4 * C) ** -0.5
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
b1, b2, gm = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
wf = R.ops._pack_mlp(w1, w2)
out, mean, rstd = torch.empty(M, C, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
code = R._lib.dtype_code
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
B = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev)


def run():
    if args.what == "fwd":
        lib.cnx_block_mlp_fwd(u.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-6, mean.data_ptr(), rstd.data_ptr(), wf.data_ptr(), b1.data_ptr(),
                              b2.data_ptr(), gm.data_ptr(), x.data_ptr(), code(x.dtype), out.data_ptr(), 0, None, M, C, S)
    elif args.what == "gemm":
        torch.mm(A, B)
    else:
        big.add_(1.0)


samples = []
stop = False


def poll():
    while not stop:
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append(json.loads(o))
        except Exception as e:                                  # noqa: BLE001
            samples.append({"error": str(e)})
        time.sleep(0.3)


for _ in range(3):
    run()
torch.cuda.synchronize()
th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < args.seconds:
    for _ in range(50):
        run()
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
res = {"what": args.what, "C": C, "launches": n, "avg_us": round(e0.elapsed_time(e1) * 1e3 / n, 1)}
keep = []
for smp in samples:
    c = smp.get("card0", smp)
    keep.append({k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk", "error"))})
res["smi"] = keep[:3] + keep[-3:]
print(json.dumps(res))
