#!/bin/bash
# GPU session 38: the whole C3 frame on a cold handle, launch grids full (default) against grids that follow the input (the default of
# sessions 20 / 21, where bench_c3.json was taken), same box
set -o pipefail
OUT=$PWD/gpurun_out/s38
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 5 --warmup 3 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
python bench.py --workload c3 --steps 2 --warmup 1 --quick --no-cpu-baseline --pixel-stride 16 > /dev/null 2>&1
scripts/ab.sh $OUT/c3_full.txt "--workload c3 --steps 1 --warmup 0 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1 ACN_GRID_PASSES=8;ACN_LEARN_GRIDS=2;ACN_LEARN_GRIDS=0"
echo done
