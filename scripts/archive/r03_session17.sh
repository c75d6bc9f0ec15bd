#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/s17
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
run() {  # dir label args env...
  local dir=$1 label=$2 args=$3; shift 3
  ( cd $dir && env "$@" timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('%-12s %-62s %9.2f ms/step  %8.1f Msamples/s  chunks %d retries %d walk_launches %d' % ('$label', '$args'[:62], d['ms_per_step'], d['value'], s['chunks'], s['retries'], s['walk_launches']))
" ) | tee -a $OUT/compare.txt
}
for rep in 1 2 3; do
for w in "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "--workload c2 --steps 10 --warmup 3 --quick"; do
  run $ROOT/old_r2 "r02" "$w" ACN_X=0
  run $ROOT "now" "$w" ACN_X=0
done
done
for w in "--workload c5 --steps 4 --warmup 2 --quick" "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16"; do
  run $ROOT/old_r2 "r02" "$w" ACN_X=0
  run $ROOT "now" "$w" ACN_X=0
done
echo "compare done" | tee -a $OUT/progress.txt
bash scripts/r03_c5bands.sh 1350 1485
