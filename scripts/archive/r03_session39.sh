#!/bin/bash
# GPU session 39: fetch batch and private-stack limit of k_walk on the two lamp scenes (k_walk is 63 / 79 % of their kernel time)
set -o pipefail
OUT=$PWD/gpurun_out/s39
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 5 --warmup 3 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
V="ACN_X=0;ACN_FETCH_WALK=32;ACN_FETCH_WALK=128;ACN_PRIVATE_LIMIT=8192;ACN_PRIVATE_LIMIT=16384;ACN_PRIVATE_LIMIT=65536;ACN_STACK_CAP=256;ACN_X=0"
scripts/ab.sh $OUT/c5.txt "--workload c5 --steps 5 --warmup 2 --quick" "$V"
scripts/ab.sh $OUT/paraffin.txt "--workload paraffin_lamp --steps 5 --warmup 2 --quick" "$V"
echo done
