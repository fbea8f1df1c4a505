cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_d; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-other-configs --cpu-steps 1 --cpu-batch 4 > $O/bench.log 2>&1
python -m pytest tests/test_gpu_ddp_rccl.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
