#!/bin/bash
# round 4, GPU session 18: whole-node scalar preloads (ACN_PRELOAD_NODES) on the small kernels (round 3: +55 % slower, the SGPR spills doubled)
set -o pipefail
OUT=$PWD/gpurun_out/s18; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
P="ACN_LIBDIR=$PWD/lib_preload"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$P"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$M;$P"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$P"
done
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M;$P"
echo session done
