#!/bin/bash
# round 4, GPU session 27: many_spheres -- a second table with the children of every compound in reverse order, for rays that run
# against the order of the first: parity, same-box A/B (ACN_NO_SC_REVERSED=1 uploads only the first), whole frame with its digest
set -o pipefail
OUT=$PWD/gpurun_out/s27; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "many_spheres or c3 or culled" > $OUT/tests_c3.log 2>&1; rc=$?; echo "pytest c3 rc $rc" | tee $OUT/progress.txt; tail -n 8 $OUT/tests_c3.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
ACN_VERBOSE=1 python bench.py $W --no-cpu-baseline 2>&1 > /dev/null | grep "simple compounds" | tee $OUT/tables.txt
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_c3.txt "$W" "ACN_NO_SC_REVERSED=1;ACN_TWO_ORDERS_ON=1"
done
echo "ab done" >> $OUT/progress.txt
timeout -k 10 400 python bench.py --workload c3 --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_c3.json > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -n 5 $OUT/bench_c3.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench_c3.json') if l.startswith('{')][-1]); print('c3 frame', d['ms_per_step'], d['value'], (d.get('frame_check') or {}).get('golden'))"
echo session done
