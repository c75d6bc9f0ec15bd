#!/bin/bash
# GPU session 16: which of this round's kernel changes costs the ~1.5 % on the small scenes (same box, three interleaved repetitions)
set -o pipefail
OUT=$PWD/gpurun_out/s16
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
run() {  # dir label args env...
  local dir=$1 label=$2 args=$3; shift 3
  ( cd $dir && env "$@" timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('%-22s %-50s %9.2f ms/step  %8.1f Msamples/s  chunks %d walk_launches %d' % ('$label', '$args'[:50], d['ms_per_step'], d['value'], s['chunks'], s['walk_launches']))
" ) | tee -a $OUT/compare.txt
}
run $ROOT warmup "--workload wine_glass_1080p --steps 8 --warmup 2 --quick" ACN_X=0 > /dev/null
for rep in 1 2 3; do
  for w in "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "--workload c2 --steps 10 --warmup 3 --quick"; do
    run $ROOT/old_r2 "r02" "$w" ACN_X=0
    run $ROOT "now" "$w" ACN_X=0
    run $ROOT "now, no prefetch" "$w" ACN_LIBDIR=$ROOT/lib_nopf
    run $ROOT "now, no early leave" "$w" ACN_LIBDIR=$ROOT/lib_noel
    run $ROOT "now, fixed passes" "$w" ACN_LEARN_PASSES=0
  done
done
echo done | tee $OUT/progress.txt
