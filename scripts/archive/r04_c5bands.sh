#!/bin/bash
# several bands of c5full in one call: scripts/r04_c5bands.sh Y0 Y1 Y2 ... (consecutive bands Y0:Y1, Y1:Y2, ...)
OUT=$PWD/gpurun_out/c5full_r04
mkdir -p $OUT
export TMPDIR=/tmp
( while true; do sleep 60; echo "alive $(date +%T)" >> $OUT/alive_bands.txt; done ) &
BEAT=$!
prev=$1; shift
for y in "$@"; do
  timeout -k 10 570 python bench.py --workload c5full --rows $prev:$y --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_rows_${prev}_$y.json > $OUT/bench_rows_${prev}_$y.json 2> $OUT/bench_rows_${prev}_$y.err
  echo "rows $prev:$y rc $?" | tee -a $OUT/bands.txt
  python - <<PY | tee -a $OUT/bands.txt
import json
try:
    d = json.load(open("$OUT/bench_rows_${prev}_$y.json")); s = d["stages"]
    print("c5full rows $prev:$y  %.3f Msamples/s  %.1f s  chunks %d retries %d ws %.1f GB sha %s" % (d["value"], d["ms_per_step"] / 1e3, s["chunks"], s["retries"], s["workspace_bytes"] / 1e9, d["frame_check"]["sha256"][:16]))
except Exception as e:
    print("no result", e)
PY
  prev=$y
done
kill $BEAT
