cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_v; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench1.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -o bench -- python3 bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_prof_g1.log 2>&1
python tools/step_breakdown.py $(ls $O/prof_g1/*kernel_trace.csv | head -1) --step 5 --top 60 --md $O/step_g1.md > $O/step_g1.log 2>&1
cp $(ls $O/prof_g1/*kernel_stats.csv | head -1) $O/kernel_stats_g1.csv
rm -rf $O/prof_g1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o k1 -- python3 tools/k1_pmc.py > $O/k1_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o k1 -- python3 tools/k1_pmc.py > $O/k1_pmc_write.log 2>&1
python3 tools/k1_traffic.py $O/pmc_fetch $O/pmc_write $O/k1_traffic.json > $O/k1_traffic.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
python tools/k1_sweep.py > $O/k1_sweep.log 2>&1
