#!/bin/bash
# GPU session 5 of round 3: round-2 tree (old_r2/) against the current one on the same box, interleaved, per workload
set -o pipefail
OUT=$PWD/gpurun_out/s5
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
echo "warm-up" | tee $OUT/progress.txt
run $ROOT warmup "--workload wine_glass_1080p --steps 8 --warmup 2 --quick" ACN_X=0 > /dev/null
for rep in 1 2; do
  for w in "--workload wine_glass_1080p --steps 8 --warmup 2 --quick" "--workload c2 --steps 8 --warmup 2 --quick" "--workload c5 --steps 4 --warmup 2 --quick" "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "--workload c3 --steps 1 --warmup 1 --quick --pixel-stride 16" "--workload c4 --steps 1 --warmup 1 --quick --pixel-stride 16" "--workload c1 --steps 20 --warmup 3 --quick"; do
    run $ROOT/old_r2 "r02" "$w" ACN_X=0
    run $ROOT "now" "$w" ACN_X=0
    run $ROOT "now, no cone cull" "$w" ACN_LIBDIR=$ROOT/lib_nocull
    run $ROOT "now, fixed 12 passes" "$w" ACN_LEARN_PASSES=0
    run $ROOT "now, k_walk w/o machine LICM" "$w" ACN_LIBDIR=$ROOT/lib_nolicm
  done
  echo "rep $rep done" | tee -a $OUT/progress.txt
done
echo done | tee -a $OUT/progress.txt
