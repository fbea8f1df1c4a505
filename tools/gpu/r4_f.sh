cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_f; mkdir -p $O
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_w5_$rep.log 2>&1
python bench.py --steps 20 --warmup 15 --no-cpu-baseline --no-other-configs > $O/bench_w15_$rep.log 2>&1
python bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-other-configs > $O/bench_w40_$rep.log 2>&1
done
