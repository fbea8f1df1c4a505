cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_m; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py -x -q > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
python tools/gemm_bench.py > $O/gemm_regstage.log 2>&1
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_glds.so python tools/gemm_bench.py > $O/gemm_glds.log 2>&1
python tools/gemm_bench.py >> $O/gemm_regstage.log 2>&1
APGD_GEMM=hip python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_hip.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto.log 2>&1
