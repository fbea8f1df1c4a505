cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_g; mkdir -p $O
for rep in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_$rep.log 2>&1
done
python bench.py --gpus 1 --steps 20 --warmup 5 --settle-steps 0 --no-cpu-baseline --no-other-configs > $O/bench_nosettle.log 2>&1
python -m pytest tests/test_gpu_ddp_rccl.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
