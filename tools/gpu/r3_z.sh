cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_z; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for rep in 1 2; do
echo "== multi on" >> $O/dw.log; python tools/dw_bench.py 256 384x14,768x7 >> $O/dw.log 2>&1
echo "== multi off" >> $O/dw.log; APGD_DW_MULTI=0 python tools/dw_bench.py 256 384x14,768x7 >> $O/dw.log 2>&1
done
echo "== half batch (attack chunks)" >> $O/dw.log
python tools/dw_bench.py 128 384x14,768x7 >> $O/dw.log 2>&1
APGD_DW_MULTI=0 python tools/dw_bench.py 128 384x14,768x7 >> $O/dw.log 2>&1
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_m1_$rep.log 2>&1
APGD_DW_MULTI=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_m0_$rep.log 2>&1
done
