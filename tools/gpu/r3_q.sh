cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_q; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py tests/test_gpu_model_ops.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for rep in 1 2; do
echo "=== spread $rep" >> $O/gemm_ab.log; python tools/gemm_bench.py >> $O/gemm_ab.log 2>&1
echo "=== burst $rep" >> $O/gemm_ab.log; APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_burst.so python tools/gemm_bench.py >> $O/gemm_ab.log 2>&1
done
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto.log 2>&1
APGD_GEMM=hip python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_hip.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg3 > $O/oc_cfg3.log 2>&1
APGD_GEMM=hip timeout 300 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg3 > $O/oc_cfg3_hip.log 2>&1
timeout 400 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg4 > $O/oc_cfg4.log 2>&1
APGD_GEMM=hip timeout 400 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg4 > $O/oc_cfg4_hip.log 2>&1
APGD_BENCH_VERBOSE=1 timeout 900 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg5 > $O/oc_cfg5.log 2>&1
