#!/bin/bash
run() {
  echo "== $*"
  make hip -B "$@" > gpurun_out/build_sweep.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build_sweep.log; return; }
  timeout -k 10 200 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  Msamples/s %.1f  ms/step %.1f  walk %.1f shade %.1f' % (d['value'], d['ms_per_step'], d['stages']['walk_ms'], d['stages']['shade_ms']))
"
}
run SHADE_WAVES=2 WALK_WAVES=2
run SHADE_WAVES=3 WALK_WAVES=2
run SHADE_WAVES=4 WALK_WAVES=3
