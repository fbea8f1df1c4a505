cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_l; mkdir -p $O
for rep in 1 2; do
for S in 2 3 4; do
APGD_ATTACK_STREAMS=$S python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_s${S}_$rep.log 2>&1
done; done
