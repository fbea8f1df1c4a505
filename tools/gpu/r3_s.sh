cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_s; mkdir -p $O
export TMPDIR=/tmp
for MODE in hip auto; do
APGD_GEMM=$MODE rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$MODE -o bench -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/bench_prof_$MODE.log 2>&1
python tools/step_breakdown.py $(ls $O/prof_$MODE/*kernel_trace.csv | head -1) --step 5 --top 70 --md $O/step_$MODE.md > $O/step_$MODE.log 2>&1
rm -rf $O/prof_$MODE
done
