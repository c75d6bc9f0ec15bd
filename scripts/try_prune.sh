#!/bin/bash
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 1000000 32; do
  echo "ACN_PRUNE_MIN=$v"
  ACN_PRUNE_MIN=$v timeout -k 10 200 python scripts/time_scene.py diamond 240 135 512 50 2>&1 | grep "iter 1" | cut -c1-50
  for w in c5 paraffin_lamp; do
    ACN_PRUNE_MIN=$v timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('$w  ms/step %.1f  walk %.1f shade %.1f hard %.1f (hard rays %.1fM)' % (d['ms_per_step'], s['walk_ms'], s['shade_ms'], s['hard_ms'], s['hard_rays'] / 1e6))
"
  done
done
scripts/quick_bench.sh c2_default
