cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_g; mkdir -p $O
APGD_GEMM_BM=128 python tools/gemm_bench.py > $O/gemm_bm128.log 2>&1
APGD_GEMM_BM=256 python tools/gemm_bench.py > $O/gemm_bm256.log 2>&1
