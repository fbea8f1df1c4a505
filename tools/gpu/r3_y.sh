cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_y; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for rep in 1 2 3; do
APGD_WGRAD_SIDE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_ws1_$rep.log 2>&1
APGD_WGRAD_SIDE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_ws0_$rep.log 2>&1
done
