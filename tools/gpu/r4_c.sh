cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_c; mkdir -p $O
python -m pytest tests/test_gpu_graph.py tests/test_gpu_ddp_rccl.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
