#!/bin/bash
# GPU session 19: parked origins of the hit machine in LDS (lib_aux) against the committed library, same box, interleaved
set -o pipefail
OUT=$PWD/gpurun_out/s19
mkdir -p $OUT
export TMPDIR=/tmp
A="ACN_LIBDIR=$PWD/actinon_amd/lib"
B="ACN_LIBDIR=$PWD/lib_aux"
echo "== parity of the variant" | tee $OUT/progress.txt
ACN_LIBDIR=$PWD/lib_aux timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --timeout 400 > $OUT/pytest_aux.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_aux.log | tee -a $OUT/progress.txt
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_c2.txt "--workload c2 --steps 10 --warmup 3 --quick" "$A;$B"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload c1 --steps 20 --warmup 3 --quick" "$A;$B"
echo done | tee -a $OUT/progress.txt
