# package power and sclk while bench.py runs its steps (polls every ~0.5 s from the start; the steps are the high plateau)
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r2_power_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "import sys,json,time; d=json.load(sys.stdin)['card0']; print(round(time.time()%1000,1), d.get('Current Socket Graphics Package Power (W)'), d.get('sclk clock speed:'))"
  sleep 0.3
done
python3 -c "import json; d=json.loads(open('gpurun_out/r2_power_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
