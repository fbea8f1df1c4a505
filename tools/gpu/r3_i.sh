cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_i; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -x -q > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
python tools/gemm_bench.py > $O/gemm_bench.log 2>&1
APGD_GEMM_BM=128 python tools/gemm_bench.py > $O/gemm_bm128.log 2>&1
APGD_GEMM_BM=256 python tools/gemm_bench.py > $O/gemm_bm256.log 2>&1
python -m pytest tests/test_gpu_configs.py -x -q -k cfg5 > $O/pytest_cfg5.log 2>&1; echo "rc=$?" >> $O/pytest_cfg5.log
