"""N > 1 on real GPUs: one process per GPU over RCCL (backend "nccl"), as ``main.py:351-359, 889-890, 1128-1152``.
Skipped on a 1-GPU box (the round's GPU box has one); the CPU twin with gloo is tests/test_ddp_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_on_one_gpu_box_refuses_two_ranks():
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs: covered by the two-rank test")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"metric"' not in r.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_bench_spawns_two_rccl_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "16", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 32 and line["scaling"] == "weak"


def test_bench_line_carries_the_contract_fields():
    """`python bench.py` on one GPU (small run): ONE JSON line with the driver's fields, the roofline object of the dominant
    kernel (both forms), a CPU baseline measured by the oracle, package power; `vs_baseline` null (no published number)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-steps", "1", "--cpu-batch", "4",
                        "--no-other-configs"], env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("adversarial images/sec") and d["unit"] == "img/s" and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "bf16"
    assert d["data"].startswith("synthetic") and "workload" in d["config"] and d["config"]["global_batch"] == 256
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "first_iter"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.3 < rf["frac"] < 1.0 and 0.3 < rf["first_iter"]["frac"] < 1.05
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "img/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["extra"]["ops_mode"] == "hip"


def test_bench_under_torch_distributed_run_with_two_ranks_sharing_the_gpu():
    """The driver's launch line for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) on the
    1-GPU box: two ranks over gloo on cuda:0 (APGD_DIST_BACKEND is the only difference from the RCCL run).  One JSON line from
    rank 0 with the whole-job rate."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(_env(), APGD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "16", "--no-cpu-baseline", "--no-other-configs"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]          # whole-job images per second


_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import revisiting_at_amd as R
rank, local, world = R.setup_distributed()
assert dist.get_backend() == os.environ.get("APGD_DIST_BACKEND", "nccl") and world == 2
dev = torch.device("cuda", local)
torch.manual_seed(0)
A = R.architecture
m = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), num_classes=10)
m.stem = A.ConvBlock1(48)
st = {"in_attack": False, "hook_in_attack": 0, "hook_total": 0}
real_attack = R.build_perturb(R.AdvConfig(attack="apgd", n_iter=2))
def attack(model, x, y):
    st["in_attack"] = True
    try:
        return real_attack(model, x, y)
    finally:
        st["in_attack"] = False
tr = R.ATTrainStep(m, "convnext_tiny", R.AdvConfig(attack="apgd", n_iter=2), dev, lr=1e-3, distributed=True, ema=True,
                   perturb=attack)
def hook(_, bucket):
    st["hook_total"] += 1
    st["hook_in_attack"] += int(st["in_attack"])
    buf = bucket.buffer().div_(world)
    dist.all_reduce(buf)                      # synchronous: gloo's CUDA work objects have no future
    fut = torch.futures.Future()
    fut.set_result(buf)
    return fut
tr.model.register_comm_hook(None, hook)
g = torch.Generator(device=dev).manual_seed(100 + rank)
x = torch.rand(4, 3, 64, 64, device=dev, generator=g)
y = torch.randint(0, 10, (4,), device=dev, generator=g)
for _ in range(2):
    tr.step(x, y)
flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
other = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(other, flat)
ok = all(torch.equal(other[0], t) for t in other)
if rank == 0:
    print(json.dumps(dict(same=bool(ok), **st)))
dist.barrier()
dist.destroy_process_group()
"""


def _run_two_ranks(tmp_path, backend):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", APGD_DIST_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, text=True))
    out = procs[0].communicate(timeout=900)[0]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_rccl_ranks_keep_identical_parameters_and_no_collective_inside_the_attack(tmp_path):
    res = _run_two_ranks(tmp_path, "nccl")
    assert res["same"] and res["hook_in_attack"] == 0 and res["hook_total"] >= 2, res


def test_two_ranks_on_one_gpu_with_the_product_model(tmp_path):
    """The 1-GPU box's stand-in for the RCCL run: two rank processes share cuda:0 over gloo (RCCL refuses two ranks per
    device).  Everything except the transport is the product path: DDP(WrappedModel(ConvNeXt)) with the fused blocks, the HIP
    attack with its gradient-sign sink inside DDP.forward, bf16 autocast, AdamW, EMA.  Parameters stay identical across the
    ranks, and no bucket reduction fires inside the attack."""
    res = _run_two_ranks(tmp_path, "gloo")
    assert res["same"] and res["hook_in_attack"] == 0 and res["hook_total"] >= 2, res
