#!/bin/bash
# GPU session 36: the whole C4 / C3 frames with the shipped binary against the committed digests (tests/golden/frame_checksums.json)
set -o pipefail
OUT=$PWD/gpurun_out/s36
mkdir -p $OUT
export TMPDIR=/tmp
for w in c4 c3; do
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
