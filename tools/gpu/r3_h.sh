cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_h; mkdir -p $O
python tools/mlp_ab.py revisiting-at_amd/libapgd_prev.so revisiting-at_amd/libapgd_hip.so --C 96,128,192,256,384 > $O/ab_prev_vs_new.log 2>&1
