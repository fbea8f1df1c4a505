cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_b; mkdir -p $O
for rep in 1 2; do
for C in 128 192 256 384; do HW=$((5376/C));
  python tools/mlp_bench.py --C $C --hw $HW --what fwd,hpre --tag g1x2 >> $O/mlp_g1x2.log 2>&1
  APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_g1x1.so python tools/mlp_bench.py --C $C --hw $HW --what fwd,hpre --tag g1x1 >> $O/mlp_g1x1.log 2>&1
done; done
