cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_t; mkdir -p $O
timeout 250 python tools/cfg_step.py vit_b 224 256 2 --dump-after 100 > $O/vit_auto.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto1.log 2>&1
APGD_GEMM=hip python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_hip1.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py tests/test_gpu_configs.py tests/test_gpu_model_ops.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
APGD_GEMM=hip timeout 250 python tools/cfg_step.py vit_b 224 256 2 --dump-after 100 > $O/vit_hip.log 2>&1
timeout 400 python tools/cfg_step.py convnext_large 320 128 3 --dump-after 200 > $O/cnxl_auto.log 2>&1
export TMPDIR=/tmp
for MODE in hip auto; do
APGD_GEMM=$MODE rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$MODE -o bench -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/bench_prof_$MODE.log 2>&1
python tools/step_breakdown.py $(ls $O/prof_$MODE/*kernel_trace.csv | head -1) --step 5 --top 70 --md $O/step_$MODE.md > $O/step_$MODE.log 2>&1
rm -rf $O/prof_$MODE
done
