cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_e; mkdir -p $O
python -m pytest tests/test_gpu_graph.py tests/test_gpu_configs.py tests/test_gpu_apgd.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_$rep.log 2>&1; done
