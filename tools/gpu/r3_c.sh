cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_c; mkdir -p $O
python -m pytest tests/test_gpu_model_ops.py -x -q -k "block or mlp or fused" > $O/pytest_blocks.log 2>&1; echo "rc=$?" >> $O/pytest_blocks.log
for rep in 1 2; do
for C in 128 192 256 384; do HW=$((5376/C));
  python tools/mlp_bench.py --C $C --hw $HW --what fwd,hpre,bwd_in --tag asm >> $O/mlp_asm.log 2>&1
  APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_gb.so python tools/mlp_bench.py --C $C --hw $HW --what fwd,hpre,bwd_in --tag builtin >> $O/mlp_builtin.log 2>&1
done; done
python tools/mlp_bench.py --C 96 --hw 56 --what fwd,bwd_in,bwd --tag asm >> $O/mlp_asm.log 2>&1
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_gb.so python tools/mlp_bench.py --C 96 --hw 56 --what fwd,bwd_in,bwd --tag builtin >> $O/mlp_builtin.log 2>&1
python tools/mlp_bench.py --C 192 --hw 28 --what bwd --tag asm >> $O/mlp_asm.log 2>&1
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_gb.so python tools/mlp_bench.py --C 192 --hw 28 --what bwd --tag builtin >> $O/mlp_builtin.log 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench.log 2>&1
