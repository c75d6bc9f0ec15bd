#!/bin/bash
# round 4, GPU session 32: many_spheres -- the table's base pointer in a VGPR pair instead of spilled SGPRs (four v_readlane per visit):
# parity of the C3 tests, same-box A/B
set -o pipefail
OUT=$PWD/gpurun_out/s32; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "many_spheres or c3 or culled" > $OUT/tests_c3.log 2>&1; rc=$?; echo "pytest c3 rc $rc" | tee $OUT/progress.txt; tail -n 4 $OUT/tests_c3.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
python bench.py $W --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_c3.txt "$W" "ACN_LIBDIR=$PWD/lib_tabs;ACN_LIBDIR=$PWD/actinon_amd/lib"
done
echo session done
