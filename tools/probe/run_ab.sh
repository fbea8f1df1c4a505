for rep in 1 2; do
for lib in hip p0; do
  export APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_$lib.so
  for cfg in "96 56 f32" "96 56 bf16" "128 56 f32" "192 28 f32" "256 28 f32" "384 14 f32"; do set -- $cfg
    python tools/mlp_bench.py --C $1 --hw $2 --resid $3 --what fwd --iters 40 --tag $lib 2>/dev/null
  done
done
done
for lib in hip p0 hip p0; do
  APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_$lib.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $lib', d['value'], d['ms_per_step'])"
done
