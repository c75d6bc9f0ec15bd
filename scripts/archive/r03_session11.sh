#!/bin/bash
# GPU session 11: lanes and workspace bound on the heavy configs
OUT=$PWD/gpurun_out/s11
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 python bench.py --steps 6 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
scripts/ab.sh $OUT/ab_c5full.txt "--workload c5full --steps 1 --warmup 0 --quick --pixel-stride 64" "ACN_X=0;ACN_LANES=2;ACN_LANES=1;ACN_LANES=1 ACN_WORKSPACE_MB=131072;ACN_WORKSPACE_MB=131072"
echo c5 done | tee $OUT/progress.txt
scripts/ab.sh $OUT/ab_c4.txt "--workload c4 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_LANES=2;ACN_LANES=1;ACN_LANES=1 ACN_WORKSPACE_MB=131072"
scripts/ab.sh $OUT/ab_c3.txt "--workload c3 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_LANES=2;ACN_LANES=1;ACN_LANES=1 ACN_WORKSPACE_MB=131072"
scripts/ab.sh $OUT/ab_s8.txt "--workload wine_glass_1080p --steps 8 --warmup 2 --quick --pixel-stride 8" "ACN_X=0;ACN_LANES=6;ACN_LANES=8;ACN_LANES=8 ACN_GRID=256 ACN_SHADE_GRID=256"
echo done | tee -a $OUT/progress.txt
