#!/bin/bash
# round 4, GPU session 17: pooled machines (the four waves of a workgroup pool their rays per root element): smoke first, then the
# suite, then same-box A/B against the build before
set -o pipefail
OUT=$PWD/gpurun_out/s17; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
tail -n 2 $OUT/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 12 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_s15"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$O;$M"
done
scripts/ab.sh $OUT/ab.txt "--workload c5full --steps 1 --warmup 1 --quick --pixel-stride 256" "$O;$M"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M"
scripts/ab.sh $OUT/ab.txt "--workload c1 --steps 20 --warmup 3 --quick" "$O;$M"
echo session done
