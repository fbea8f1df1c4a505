cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_x; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench1.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
