#!/bin/bash
# round 4, GPU session 19: the lock-step machine's ray origin parked in LDS (OrgLds): suite + same-box A/B + traffic of k_walk
set -o pipefail
OUT=$PWD/gpurun_out/s19; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 12 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_nopark"
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$O;$M"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$O;$M"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$O;$M"
done
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/pmc_$c; mkdir -p $d
  rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 0 --quick --no-cpu-baseline > $d/log.txt 2>&1
  python3 scripts/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) > $OUT/pmc_$c.txt
  find $d -name "*.csv" -size +5M -delete
done
grep "k_walk\|k_hard" $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt | cut -c1-140
echo session done
