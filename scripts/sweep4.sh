#!/bin/bash
run() {
  echo "== $*"
  make hip -B "$@" > gpurun_out/build_sweep.log 2>&1 || { echo "build failed"; grep error gpurun_out/build_sweep.log | head -3; return; }
  timeout -k 10 200 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('  Msamples/s %.1f  ms/step %.1f  walk %.1f shade %.1f hard %.1f' % (d['value'], d['ms_per_step'], s['walk_ms'], s['shade_ms'], s['hard_ms']))
"
}
run EXTRA_DEFS="-DACN_SIDE_INLINE=0 -DACN_HIT_INLINE=0"
run EXTRA_DEFS="-DACN_SIDE_INLINE=1 -DACN_HIT_INLINE=0"
run EXTRA_DEFS="-DACN_SIDE_INLINE=1 -DACN_HIT_INLINE=1"
run EXTRA_DEFS="-DACN_SIDE_INLINE=1 -DACN_HIT_INLINE=1" WALK_WAVES=3
run EXTRA_DEFS="-DACN_SIDE_INLINE=1 -DACN_HIT_INLINE=0" WALK_WAVES=3
