export APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_abl.so
for d in 0 1 2 32 3 34 35; do
  APGD_BLK_DBG=$d python tools/blk_trace.py --C 384 --hw 14 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dbg $d', d['event_us'], d['phases_us']['hidden loop']['median'], d['shader_clock_GHz_in_the_hidden_loop'], d['hidden_loop_cycles_per_slice_split'])"
done
