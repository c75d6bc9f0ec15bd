#!/bin/bash
# GPU box: rebuild libactinon_hip.so with different launch-bound occupancies and time bench workload c2
for cfg in "4 2" "3 2" "2 2" "2 1"; do
  set -- $cfg
  make hip SHADE_WAVES=$1 WALK_WAVES=$2 -B > gpurun_out/build_$1_$2.log 2>&1 || { echo "build failed $cfg"; continue; }
  echo "== shade_waves=$1 walk_waves=$2"
  timeout -k 10 200 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  Msamples/s %.1f  ms/step %.1f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stages'].items()}))
"
done
