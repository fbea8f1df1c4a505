cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_f; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hip -o step -- python3 bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/prof_hip.log 2>&1
python tools/step_breakdown.py $O/prof_hip/step_kernel_trace.csv --step 3 > $O/step_hip.md 2>&1
APGD_GEMM=lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lib -o step -- python3 bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/prof_lib.log 2>&1
python tools/step_breakdown.py $O/prof_lib/step_kernel_trace.csv --step 3 > $O/step_lib.md 2>&1
cp $O/prof_hip/step_kernel_stats.csv $O/step_hip_kernel_stats.csv; cp $O/prof_lib/step_kernel_stats.csv $O/step_lib_kernel_stats.csv
rm -rf $O/prof_hip $O/prof_lib
for rep in 1 2; do for C in 192 384; do HW=$((5376/C)); python tools/mlp_bench.py --C $C --hw $HW --what fwd,hpre >> $O/mlp.log 2>&1; done; done
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_hip.log 2>&1
APGD_GEMM=lib python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_lib.log 2>&1
