#!/bin/bash
# round 4, GPU session 31: the element indices of the root loops requested one ahead (ElemAhead): suite, same-box A/B
set -o pipefail
OUT=$PWD/gpurun_out/s31; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 8 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
V="ACN_LIBDIR=$PWD/lib_noahead;ACN_LIBDIR=$PWD/actinon_amd/lib"
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
done
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
echo session done
