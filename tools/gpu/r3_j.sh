cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_j; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -x -q > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
python tools/gemm_bench.py > $O/gemm_bench.log 2>&1
python tools/mlp_ab.py revisiting-at_amd/libapgd_prev.so revisiting-at_amd/libapgd_hip.so --C 96,192,384 > $O/ab_prologue.log 2>&1
python -m pytest tests/test_gpu_configs.py -x -q -k cfg5 > $O/pytest_cfg5.log 2>&1; echo "rc=$?" >> $O/pytest_cfg5.log
python -m pytest tests/test_gpu_model_ops.py -x -q -k "block or mlp or fused" > $O/pytest_blocks.log 2>&1; echo "rc=$?" >> $O/pytest_blocks.log
