#!/bin/bash
# GPU session 31: k_walk compiled with -mllvm -amdgpu-sched-strategy=max-ilp (lib_ilp) against the committed library
set -o pipefail
OUT=$PWD/gpurun_out/s31
mkdir -p $OUT
export TMPDIR=/tmp
A="ACN_LIBDIR=$PWD/actinon_amd/lib"
B="ACN_LIBDIR=$PWD/lib_ilp"
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
ACN_LIBDIR=$PWD/lib_ilp timeout -k 10 300 python bench.py --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/digest.json 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('digest of the 1080p frame with the variant:', json.loads(l)['frame_check']['golden'])" | tee $OUT/progress.txt
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_c2.txt "--workload c2 --steps 10 --warmup 3 --quick" "$A;$B"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$A;$B"
echo done
