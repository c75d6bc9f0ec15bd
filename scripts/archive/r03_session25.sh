#!/bin/bash
# GPU session 25: the 1/8 share of the 1080p frame against the private-stack limit and the number of walk passes
set -o pipefail
OUT=$PWD/gpurun_out/s25
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
A="--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8"
scripts/ab.sh $OUT/stride8.txt "$A" "ACN_X=0;ACN_PRIVATE_LIMIT=8192;ACN_PRIVATE_LIMIT=65536;ACN_PRIVATE_LIMIT=131072;ACN_PRIVATE_LIMIT=262144;ACN_PRIVATE_LIMIT=1048576"
scripts/ab.sh $OUT/stride8.txt "$A" "ACN_X=0;ACN_WALK_PASSES=3;ACN_WALK_PASSES=5;ACN_WALK_PASSES=8;ACN_LANES=3;ACN_LANES=5;ACN_LANES=6"
scripts/ab.sh $OUT/stride8.txt "$A" "ACN_X=0;ACN_WALK_PASSES=3 ACN_PRIVATE_LIMIT=131072;ACN_WALK_PASSES=5 ACN_PRIVATE_LIMIT=131072;ACN_WALK_GRID=1024;ACN_WALK_GRID=256;ACN_STACK_CAP=1024 ACN_PRIVATE_LIMIT=262144"
echo done
