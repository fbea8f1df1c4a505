cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_j; mkdir -p $O
for rep in 1 2 3; do
APGD_BLK_FWD_RES=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res0_$rep.log 2>&1
APGD_BLK_FWD_RES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res1_$rep.log 2>&1
done
APGD_BLK_FWD_RES=1 APGD_ATTACK_STREAMS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res1_s1.log 2>&1
APGD_BLK_FWD_RES=0 APGD_ATTACK_STREAMS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res0_s1.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
