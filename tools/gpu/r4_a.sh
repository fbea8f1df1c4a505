cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_a; mkdir -p $O
python -m pytest tests/test_gpu_model_ops.py tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python tools/dw_bench.py 256 384x14,768x7 > $O/dw.log 2>&1
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_$rep.log 2>&1
done
