cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_u; mkdir -p $O
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto1.log 2>&1
APGD_GEMM_AUTO_MAX=0 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_automax0.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench_auto2.log 2>&1
timeout 250 python tools/cfg_step.py vit_b 224 256 2 --dump-after 100 > $O/vit_auto.log 2>&1
timeout 400 python tools/cfg_step.py convnext_large 320 128 3 --dump-after 200 > $O/cnxl_auto.log 2>&1
( time timeout 1200 python bench.py ) > $O/bench_default.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
