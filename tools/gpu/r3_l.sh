cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_l; mkdir -p $O
python tools/k1_ab.py > $O/k1_ab.log 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $O/bench.log 2>&1
