#!/bin/bash
# GPU session 37: the whole C3 / C4 frames with the shipped binary on a warmed-up box (cold handle, as bench_c3.json / bench_c4.json)
set -o pipefail
OUT=$PWD/gpurun_out/s37
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 5 --warmup 3 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
python bench.py --workload c3 --steps 2 --warmup 1 --quick --no-cpu-baseline --pixel-stride 16 > /dev/null 2>&1
for w in c3 c4; do
  timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_$w.json > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { tail -n 3 $OUT/bench_$w.err; exit 1; }
  python3 - $OUT/bench_$w.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']
        print("%s whole frame: %.1f ms  %.2f Msamples/s  chunks %d retries %d  digest %s" % (d['config']['workload'][:24], d['ms_per_step'], d['value'], s['chunks'], s['retries'], d['frame_check'].get('golden')))
PY
done
echo done
