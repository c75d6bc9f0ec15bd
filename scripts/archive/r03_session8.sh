#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/s8
mkdir -p $OUT
B="--workload wine_glass_1080p --steps 10 --warmup 3 --quick"
scripts/ab.sh $OUT/ab.txt "$B" "ACN_X=0;ACN_LIBDIR=$PWD/lib_hp2;ACN_X=1;ACN_LIBDIR=$PWD/lib_hp2 ACN_X=1"
scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 4 --warmup 2 --quick" "ACN_X=0;ACN_LIBDIR=$PWD/lib_hp2"
scripts/ab.sh $OUT/ab_c4.txt "--workload c4 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_LIBDIR=$PWD/lib_hp2"
bash scripts/r03_c5band.sh 0 1080 --cpu-window 240x135
