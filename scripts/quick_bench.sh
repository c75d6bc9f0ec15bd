#!/bin/bash
# usage: scripts/quick_bench.sh [label] ; env vars pass through
timeout -k 10 200 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('$1  Msamples/s %.1f  ms/step %.1f  walk %.1f shade %.1f hard %.1f (hard rays %.1fM)' % (d['value'], d['ms_per_step'], s['walk_ms'], s['shade_ms'], s['hard_ms'], s['hard_rays'] / 1e6))
"
