cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_o; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -o bench -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs --graph 1 > $O/bench_prof_g1.log 2>&1
python tools/step_breakdown.py $(ls $O/prof_g1/*kernel_trace.csv | head -1) --step 5 --top 60 --md $O/step_g1.md > $O/step_g1.log 2>&1
cp $(ls $O/prof_g1/*kernel_stats.csv | head -1) $O/kernel_stats_g1.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g0 -o bench -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/bench_prof_g0.log 2>&1
python tools/step_breakdown.py $(ls $O/prof_g0/*kernel_trace.csv | head -1) --step 5 --top 60 --md $O/step_g0.md > $O/step_g0.log 2>&1
cp $(ls $O/prof_g0/*kernel_stats.csv | head -1) $O/kernel_stats_g0.csv
rm -rf $O/prof_g1 $O/prof_g0
python bench.py --steps 10 --warmup 4 > $O/bench_full.log 2>&1
