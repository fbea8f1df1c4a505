set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python tools/conv2_bench.py 48 256 112 > $O/conv2_w12.log 2>&1
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_c2w8.so python tools/conv2_bench.py 48 256 112 > $O/conv2_w8.log 2>&1
python tools/conv2_bench.py 48 256 112 >> $O/conv2_w12.log 2>&1
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_c2w8.so python tools/conv2_bench.py 48 256 112 >> $O/conv2_w8.log 2>&1
python tools/dw_bench.py 256 > $O/dw.log 2>&1
for C in 96 192 384; do HW=$((5376/C)); python tools/mlp_bench.py --C $C --hw $HW --what fwd,bwd_in,hpre,bwd >> $O/mlp.log 2>&1; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench.log 2>&1
