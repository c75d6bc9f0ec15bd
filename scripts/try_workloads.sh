#!/bin/bash
for w in c1 c4 c3; do
  echo "== $w"
  timeout -k 10 280 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('  %s %.1f  ms/step %.1f  walk %.1f shade %.1f hard %.1f chunks %d retries %d levels %d' % (d['unit'], d['value'], d['ms_per_step'], s['walk_ms'], s['shade_ms'], s['hard_ms'], s['chunks'], s['retries'], s['levels']))
    elif 'rror' in l: print(l.strip()[:300])
"
done
