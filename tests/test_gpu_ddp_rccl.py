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
    # (algorithmic bytes are counted at fp32 - with int8 gradient signs the fused update + row-move kernel moves fewer: frac may pass 1;
    #  the physical reading is frac_moved)
    assert 0.3 < rf["frac"] < 1.3 and 0.3 < rf["first_iter"]["frac"] < 1.3 and 0.3 < rf["frac_moved"] < 1.0
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
m = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), num_classes=12)
m.stem = A.ConvBlock1(48)
st = {"in_attack": False, "hook_in_attack": 0, "hook_total": 0}
real_attack = R.build_perturb(R.AdvConfig(attack="apgd", n_iter=2))
def attack(model, x, y):
    st["in_attack"] = True
    try:
        return real_attack(model, x, y)
    finally:
        st["in_attack"] = False
GS = os.environ.get("APGD_TEST_GRAD_SYNC", "flat")
for name in ("all_reduce",):
    real = getattr(dist, name)
    def counted(*a, _real=real, **kw):
        st["hook_total"] += 1
        st["hook_in_attack"] += int(st["in_attack"])
        return _real(*a, **kw)
    setattr(dist, name, counted)
tr = R.ATTrainStep(m, "convnext_tiny", R.AdvConfig(attack="apgd", n_iter=2, graph=1 if GS == "flat" else 0), dev, lr=1e-3,
                   distributed=True, ema=True, perturb=attack if GS == "ddp" else None, grad_sync=GS)
FAIL_RANK = int(os.environ.get("APGD_TEST_FAIL_CAPTURE_RANK", "-1"))
if rank == FAIL_RANK:
    # this rank's training-pass capture is refused: it runs the pass from Python while the other rank replays its three segments
    class _Refused:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused (test)")
    tr._graph_cls = _Refused
if GS == "ddp":
    def hook(_, bucket):
        st["hook_total"] += 1
        st["hook_in_attack"] += int(st["in_attack"])
        buf = bucket.buffer().div_(world)
        dist.all_reduce(buf)                      # synchronous: gloo's CUDA work objects have no future
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut
    tr.model.register_comm_hook(None, hook)
else:
    # the product path of an N > 1 rank: FlatGradSync + graphs.  The attack (adv.graph=1) is replayed from hipGraphs, which cannot
    # contain a Python-level collective; `in_attack` brackets the perturb call of every step
    real_perturb = tr.inner.perturb
    def bracketed(model, x, y):
        st["in_attack"] = True
        try:
            return real_perturb(model, x, y)
        finally:
            st["in_attack"] = False
    tr.inner.perturb = bracketed
g = torch.Generator(device=dev).manual_seed(100 + rank)
x = torch.rand(4, 3, 64, 64, device=dev, generator=g)
y = torch.randint(0, 10, (4,), device=dev, generator=g)
n_steps = 2 if GS == "ddp" else 6
for _ in range(n_steps):
    tr.step(x, y)
if GS == "flat":
    st["train_graph_segments"] = [v.n_graphs for v in tr._tg.values() if v is not None]
    st["reduces"] = tr.sync.reduces
    st["attack_replays"] = R.graphed.STATS["replays"]
    mine = dict(rank=rank, segments=st["train_graph_segments"], failed=sum(v is None for v in tr._tg.values()), reduces=tr.sync.reduces)
    alls = [None] * world
    dist.all_gather_object(alls, mine)
    st["ranks"] = alls
flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
other = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(other, flat)
ok = all(torch.equal(other[0], t) for t in other)
if rank == 0:
    print(json.dumps(dict(same=bool(ok), **st)))
dist.barrier()
dist.destroy_process_group()
"""


def _run_two_ranks(tmp_path, backend, grad_sync="flat", fail_rank=None):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", APGD_DIST_BACKEND=backend, APGD_TEST_GRAD_SYNC=grad_sync,
                   APGD_TEST_FAIL_CAPTURE_RANK=str(-1 if fail_rank is None else fail_rank))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, text=True))
    out = procs[0].communicate(timeout=900)[0]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return json.loads(out.strip().splitlines()[-1])


def _check_flat(res):
    # six steps: three eager, then the training pass captured as THREE graph segments (forward + late backward | early backward |
    # AdamW + EMA) with the two all-reduces between them, replayed from then on; two exchanges per step, none inside the attack
    assert res["same"] and res["hook_in_attack"] == 0, res
    assert res["train_graph_segments"] == [3], res
    assert res["reduces"] == 12 and res["hook_total"] >= 12 and res["attack_replays"] >= 3, res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_rccl_ranks_keep_identical_parameters_and_no_collective_inside_the_attack(tmp_path):
    _check_flat(_run_two_ranks(tmp_path, "nccl"))


def test_two_ranks_on_one_gpu_with_the_product_model(tmp_path):
    """The 1-GPU box's stand-in for the RCCL run: two rank processes share cuda:0 over gloo (RCCL refuses two ranks per
    device).  Everything except the transport is the product path of an N > 1 rank: WrappedModel(ConvNeXt) with the fused blocks,
    the HIP attack replayed from hipGraphs, bf16 autocast, the training pass replayed from three graph segments with
    ``FlatGradSync``'s all-reduces between them, capturable AdamW, EMA.  Parameters stay identical across the ranks."""
    _check_flat(_run_two_ranks(tmp_path, "gloo"))


def test_two_ranks_on_one_gpu_one_of_them_without_its_training_graph(tmp_path):
    """Round 6 (the device twin of tests/test_ddp_gloo.py::test_a_rank_whose_training_pass_capture_failed_*): rank 1's capture of the
    training pass is refused, so it runs the pass from Python (real kernels, eager) while rank 0 replays its three hipGraph segments -
    both issue the flat path's two all-reduces per step between the same points of the pass: no hang, 12 exchanges in 6 steps on each,
    identical parameters."""
    res = _run_two_ranks(tmp_path, "gloo", fail_rank=1)
    assert res["same"] and res["hook_in_attack"] == 0, res
    by = {r["rank"]: r for r in res["ranks"]}
    assert by[0]["segments"] == [3] and by[0]["failed"] == 0, res
    assert by[1]["segments"] == [] and by[1]["failed"] == 1, res
    assert by[0]["reduces"] == 12 and by[1]["reduces"] == 12, res


def test_two_ranks_on_one_gpu_with_torch_ddp(tmp_path):
    """``grad_sync="ddp"`` (``main.py:889-890`` literally: DistributedDataParallel around the wrapped model, eager training pass):
    identical parameters, and no bucket reduction fires inside the attack, which runs inside DDP.forward."""
    res = _run_two_ranks(tmp_path, "gloo", "ddp")
    assert res["same"] and res["hook_in_attack"] == 0 and res["hook_total"] >= 2, res


def test_flat_gradient_path_on_one_gpu_is_the_same_step_and_costs_the_host_little():
    """``bench.py --ddp-path 1`` on one GPU: the N > 1 code (two-call backward into flat buffers, three graph segments; the
    collectives are no-ops without a process group) trains like the one-graph step (to the run-to-run noise of the library's
    filter-gradient kernels), and a replayed step() call stays a handful of launches: <= 5 ms of host time."""
    import time
    import revisiting_at_amd as R
    R._lib.load()

    def run(gs):
        R.graphed.reset()
        torch.manual_seed(5)
        A = R.architecture
        m = A.ConvNeXt(depths=(1, 1, 1, 1), dims=(96, 192, 384, 768), num_classes=12)
        m.stem = A.ConvBlock1(48)
        tr = R.ATTrainStep(m, "convnext_tiny", R.AdvConfig(attack="apgd", n_iter=2, eps=4 / 255, graph=1), "cuda", lr=1e-3,
                           amp_dtype=torch.bfloat16, ema=True, ema_decay=0.9, grad_sync=gs)
        g = torch.Generator(device="cuda").manual_seed(9)
        losses, host = [], []
        for i in range(8):
            x = torch.rand(4, 3, 64, 64, device="cuda", generator=g)
            y = torch.randint(0, 12, (4,), device="cuda", generator=g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = tr.step(x, y)
            host.append(time.perf_counter() - t0)
            losses.append(float(loss))
        segs = [v.n_graphs for v in tr._tg.values() if v is not None]
        return losses, [p.detach().clone() for p in tr.inner.parameters()], segs, host
    l1, p1, s1, h1 = run("flat")
    l0, p0, s0, h0 = run(None)
    assert s1 == [3] and s0 == [1], (s1, s0)
    assert max(abs(a - b) for a, b in zip(l1, l0)) <= 2e-3 * max(abs(v) for v in l0), (l1, l0)
    num = sum(float((a.float() - b.float()).pow(2).sum()) for a, b in zip(p1, p0))
    den = sum(float(b.float().pow(2).sum()) for b in p0)
    assert (num / den) ** 0.5 <= 1e-3, (num / den) ** 0.5
    assert min(h1[5:]) <= 5e-3, h1
