for k in 0 1 2 3 5 8; do
  d=$((16 + k * 256))
  for C in 96 192; do
    hw=56; [ $C = 192 ] && hw=28
    APGD_BLK_DBG=$d APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_abl.so python tools/mlp_bench.py --C $C --hw $hw --what fwd --tag stagger$k 2>/dev/null
  done
done
