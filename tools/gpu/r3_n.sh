cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_n; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q > $O/pytest_graph.log 2>&1; echo "rc=$?" >> $O/pytest_graph.log
for rep in 1 2; do
APGD_ATTACK_STREAMS=2 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_s2_$rep.log 2>&1
APGD_ATTACK_STREAMS=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_s1_$rep.log 2>&1
done
APGD_ATTACK_STREAMS=3 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_s3_1.log 2>&1
