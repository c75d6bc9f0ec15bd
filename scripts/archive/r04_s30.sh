#!/bin/bash
# round 4, GPU session 30: the pruning descent behind a real call (one copy per kernel: the library shrinks by 24 %) and at depth 2,
# both with the prune levels of session 29: same-box A/B
set -o pipefail
OUT=$PWD/gpurun_out/s30; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
V="ACN_LIBDIR=$PWD/actinon_amd/lib;ACN_LIBDIR=$PWD/lib_pcall;ACN_LIBDIR=$PWD/lib_pd2"
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
done
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
ACN_LIBDIR=$PWD/lib_pcall timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "parity or arrangement" > $OUT/tests_pcall.log 2>&1; echo "pytest pcall rc $?"; tail -n 3 $OUT/tests_pcall.log
echo session done
