for rep in 1 2; do
for h in 384 none; do
  APGD_BLOCK_HPRE=$h python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench hpre=$h', d['value'], d['ms_per_step'], d['extra']['attack_only_img_s'], d['extra']['package_power'])"
done
done
