#!/bin/bash
# GPU session 34: max-ILP scheduling for k_shade and the hard-ray kernels as well (lib_ilp2) against the committed library
set -o pipefail
OUT=$PWD/gpurun_out/s34
mkdir -p $OUT
export TMPDIR=/tmp
A="ACN_LIBDIR=$PWD/actinon_amd/lib"
B="ACN_LIBDIR=$PWD/lib_ilp2"
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
ACN_LIBDIR=$PWD/lib_ilp2 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/digest.json 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('digest of the 1080p frame with the variant:', json.loads(l)['frame_check']['golden'])" | tee $OUT/progress.txt
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 6 --warmup 2 --quick" "$A;$B"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
done
echo done
