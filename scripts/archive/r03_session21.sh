#!/bin/bash
# GPU session 21: the fill controller of the chunk planner on the whole C3 / C4 frames (cold handle: one frame, no warm-up)
set -o pipefail
OUT=$PWD/gpurun_out/s21
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for cfg in "4 0.85" "16 0.85" "16 0.93" "64 0.93"; do
  set -- $cfg
  for w in c4 c3; do
    ACN_DEBUG_CHUNKS=1 ACN_FILL_RECOVER=$1 ACN_RATE_DECAY=$2 timeout -k 10 200 python bench.py --workload $w --steps 1 --warmup 0 --quick --no-cpu-baseline > $OUT/${w}_$1_$2.json 2> $OUT/${w}_$1_$2.err || { tail -n 3 $OUT/${w}_$1_$2.err; exit 1; }
    python - $OUT/${w}_$1_$2.json "$w recover $1 decay $2" <<'PY' | tee -a $OUT/summary.txt
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']
        print("%-28s %9.1f ms  %6.2f Msamples/s  chunks %d retries %d" % (sys.argv[2], d['ms_per_step'], d['value'], s['chunks'], s['retries']))
PY
  done
done
echo done
