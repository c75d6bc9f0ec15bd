#!/bin/bash
# GPU session 9 of round 3: heavy configs, round-2 tree / current / current without reservation prefetch / uniform workspace; then
# the whole frame of BASELINE.json configs[4]
set -o pipefail
OUT=$PWD/gpurun_out/s9
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
run() {  # dir label args env...
  local dir=$1 label=$2 args=$3; shift 3
  ( cd $dir && env "$@" timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('%-26s %-34s %9.2f ms/step  %8.1f Msamples/s  chunks %d retries %d walk_launches %d walk_steps %d' % ('$label', '$args'[:34], d['ms_per_step'], d['value'], s['chunks'], s['retries'], s['walk_launches'], s['walk_steps']))
" ) | tee -a $OUT/compare.txt
}
run $ROOT warmup "--workload wine_glass_1080p --steps 8 --warmup 2 --quick" ACN_X=0 > /dev/null
for rep in 1 2; do
  for w in "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "--workload c5 --steps 4 --warmup 2 --quick" "--workload paraffin_lamp --steps 4 --warmup 2 --quick"; do
    run $ROOT/old_r2 "r02" "$w" ACN_X=0
    run $ROOT "now" "$w" ACN_X=0
    run $ROOT "now, no prefetch" "$w" ACN_LIBDIR=$ROOT/lib_nopf
    run $ROOT "now, uniform workspace" "$w" ACN_WS_UNIFORM=1
  done
  echo "rep $rep done" | tee -a $OUT/progress.txt
done
mkdir -p $PWD/gpurun_out/c5full
( while true; do sleep 60; echo "alive $(date +%T)" >> $OUT/alive.txt; done ) &
BEAT=$!
timeout -k 10 700 python bench.py --workload c5full --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $PWD/gpurun_out/c5full/checksum_full.json > $PWD/gpurun_out/c5full/bench_full.json 2> $PWD/gpurun_out/c5full/bench_full.err
echo "c5full rc $?" | tee -a $OUT/progress.txt
kill $BEAT
python - <<'PY' | tee -a $OUT/progress.txt
import json
d = json.load(open("gpurun_out/c5full/bench_full.json")); s = d["stages"]
print("c5full %.3f Msamples/s  %.1f s  chunks %d retries %d ws %.1f GB sha %s" % (d["value"], d["ms_per_step"] / 1e3, s["chunks"], s["retries"], s["workspace_bytes"] / 1e9, d["frame_check"]["sha256"][:16]))
PY
echo done | tee -a $OUT/progress.txt
