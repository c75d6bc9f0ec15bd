#!/bin/bash
# round 4, GPU session 23: many_spheres (C3) -- entries behind the best hit are skipped where the envelope is verified to bound its
# subtree (CULL): parity (incl. culled = plain to the bit), same-box A/B, whole frame with its digest
set -o pipefail
OUT=$PWD/gpurun_out/s23; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "many_spheres or c3 or culled" > $OUT/tests_c3.log 2>&1; rc=$?; echo "pytest c3 rc $rc" | tee $OUT/progress.txt; tail -n 8 $OUT/tests_c3.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
ACN_VERBOSE=1 python bench.py $W --no-cpu-baseline 2>&1 > /dev/null | grep "simple compounds" | tee $OUT/bounding.txt   # warms the box, too
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_sc_nocull"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_c3.txt "$W" "$O;$M"
done
echo "ab done" >> $OUT/progress.txt
timeout -k 10 400 python bench.py --workload c3 --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_c3.json > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -n 5 $OUT/bench_c3.err; exit 1; }
cut -c1-200 $OUT/bench_c3.json; cat $OUT/checksum_c3.json | cut -c1-300
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/progress.txt; tail -n 8 $OUT/tests_gpu.log
echo session done
