python -m pytest tests/test_gpu_model_ops.py -q -x -k "fused or block" 2>&1 | tail -2
for cfg in "96 56" "128 56" "192 28" "256 28" "384 14"; do set -- $cfg
  python tools/mlp_bench.py --C $1 --hw $2 --what fwd --tag pipe 2>/dev/null
done
export APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_abl.so
python tools/blk_trace.py --C 96 --hw 56 2>/dev/null
python tools/blk_trace.py --C 192 --hw 28 2>/dev/null
