cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_w; mkdir -p $O
python -m pytest tests/test_gpu_graph.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench1.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench2.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench3.log 2>&1
