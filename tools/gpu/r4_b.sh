cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_b; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_soak.log 2>&1
( time timeout 1200 python bench.py ) > $O/bench_default.log 2>&1
