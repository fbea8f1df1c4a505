for n in 0 4 8; do
  for C in 96 192; do
    hw=56; [ $C = 192 ] && hw=28
    APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_np$n.so python tools/mlp_bench.py --C $C --hw $hw --what fwd --tag np$n
  done
done
