cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_k; mkdir -p $O
for rep in 1 2; do
for V in 0 1; do
APGD_BLK_FWD_RES=$V python tools/mlp_bench.py --C 96 --hw 56 --what fwd --tag res$V >> $O/mlp96.log 2>&1
done; done
APGD_BLK_FWD_RES=1 python -m pytest tests/test_gpu_model_ops.py -x -q -k "block or mlp or fused" > $O/pytest_res1.log 2>&1; echo "rc=$?" >> $O/pytest_res1.log
for rep in 1 2 3; do
APGD_BLK_FWD_RES=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res0_$rep.log 2>&1
APGD_BLK_FWD_RES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_res1_$rep.log 2>&1
done
