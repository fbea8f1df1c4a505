for rep in 1 2; do
for lib in hip c1; do
  export APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_$lib.so
  for cfg in "96 56" "128 56" "192 28" "256 28"; do set -- $cfg
    python tools/mlp_bench.py --C $1 --hw $2 --what bwd_in --iters 30 --tag $lib 2>/dev/null
  done
done
done
unset APGD_HIP_LIB
python -m pytest tests/test_gpu_model_ops.py -q -x -k "fused or block" 2>&1 | tail -2
