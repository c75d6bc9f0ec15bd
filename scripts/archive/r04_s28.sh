#!/bin/bash
# round 4, GPU session 28: how deep surely_outside descends before a ray is handed to a machine (ACN_PRUNE_DEPTH 1 / 2 / 3 = default / 5):
# same-box A/B on every workload
set -o pipefail
OUT=$PWD/gpurun_out/s28; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
V="ACN_LIBDIR=$PWD/lib_pd1;ACN_LIBDIR=$PWD/lib_pd2;ACN_LIBDIR=$PWD/actinon_amd/lib;ACN_LIBDIR=$PWD/lib_pd5"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
done
scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
echo session done
