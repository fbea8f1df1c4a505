cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_e; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -x -q > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
python -m pytest tests/test_gpu_graph.py -x -q > $O/pytest_graph.log 2>&1; echo "rc=$?" >> $O/pytest_graph.log
python tools/gemm_bench.py > $O/gemm_bench.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_hip.log 2>&1
APGD_GEMM=lib python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_lib.log 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
