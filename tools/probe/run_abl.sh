for m in 14 30 16; do
  for cfg in "384 14" "192 28"; do set -- $cfg
  APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_pa$m.so python tools/blk_trace.py --C $1 --hw $2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PIPE_ABL $m C', d['C'], 'event', d['event_us'], 'loop_us', d['phases_us']['hidden loop']['median'], 'GHz', d['shader_clock_GHz_in_the_hidden_loop'], d['hidden_loop_cycles_per_slice_split'])"
  done
done
