"""bench.py's own N-rank launcher (``spawn_ranks``, the reference's ``mp.spawn`` of main.py:1128-1152) driven on the CPU with a
gloo worker: 8 ranks rendezvous on 127.0.0.1, all-reduce, rank 0's JSON line is relayed; a rank that dies early ends the job at
once with a non-zero status (round-2 advice: the parent used to block on rank 0's pipe while rank 0 sat in a collective)."""
import argparse
import json
import os
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = textwrap.dedent("""
    import json, os, sys, time
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    if mode == "die" and rank == 3:
        sys.stderr.write("rank 3: simulated start-up failure\\n")
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((4,), float(rank + 1))
    dist.all_reduce(t)
    assert float(t[0]) == world * (world + 1) / 2
    assert int(os.environ["OMP_NUM_THREADS"]) >= 1 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if mode == "die":
        time.sleep(600)                       # the survivors sit in "a collective"
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test", "n_gpus": world, "sum": float(t[0])}), flush=True)
    dist.destroy_process_group()
""")


def _run(tmp_path, capsys, mode, n=8):
    import bench
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    t0 = time.monotonic()
    rc = bench.spawn_ranks(argparse.Namespace(gpus=n), child_argv=[sys.executable, str(w), mode], ndev=n)
    return rc, time.monotonic() - t0, capsys.readouterr()


def test_eight_gloo_ranks_through_the_launcher(tmp_path, capsys):
    rc, dt, io = _run(tmp_path, capsys, "ok")
    assert rc == 0, io.err
    line = json.loads([ln for ln in io.out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["sum"] == 36.0


def test_a_dead_rank_ends_the_job_instead_of_hanging_it(tmp_path, capsys):
    rc, dt, io = _run(tmp_path, capsys, "die")
    assert rc == 1
    assert dt < 120, f"the launcher waited {dt:.0f} s for ranks parked behind a dead one"
    assert "rank 3" in io.err and "simulated start-up failure" in io.err
    assert "rank exit codes" in io.err


def test_fewer_devices_than_ranks_is_refused(tmp_path, capsys):
    import bench
    assert bench.spawn_ranks(argparse.Namespace(gpus=4), child_argv=[sys.executable, "-c", "pass"], ndev=1) == 2


def test_a_failed_graph_capture_is_an_error_entry_not_a_silent_eager_measurement():
    """bench.py --strict fails on extra.errors; a training pass (or attack) that fell back to its eager form although replay was
    asked for must show up there (round-4 review: under RCCL a failed three-segment capture silently benchmarked the eager pass)."""
    import bench
    ok = {"train_graph": {"enabled": True, "captured": 1, "failed": 0, "segments": [3]}, "attack_graph": {"captures": 1, "failed": 0}}
    assert "error" not in bench.mark_graph_fallbacks(ok, True, True)["train_graph"] and "error" not in ok["attack_graph"]
    bad = {"train_graph": {"enabled": True, "captured": 0, "failed": 1, "segments": []}, "attack_graph": {"captures": 1, "failed": 0}}
    bench.mark_graph_fallbacks(bad, True, True)
    assert "error" in bad["train_graph"] and "error" not in bad["attack_graph"]
    never = {"train_graph": {"enabled": True, "captured": 0, "failed": 0}, "attack_graph": {"captures": 0, "failed": 0}}
    bench.mark_graph_fallbacks(never, True, True)
    assert "error" in never["train_graph"] and "error" in never["attack_graph"]
    eager = {"train_graph": {"enabled": False, "captured": 0, "failed": 0}, "attack_graph": {"captures": 0, "failed": 0}}
    bench.mark_graph_fallbacks(eager, False, False)                      # --graph 0: nothing was asked for, nothing is wrong
    assert "error" not in eager["train_graph"] and "error" not in eager["attack_graph"]
