#!/bin/bash
# round 4, GPU session 16: where the waves of the machine kernels spend their time on the lamp scenes and the diamond (phase timers)
set -o pipefail
OUT=$PWD/gpurun_out/s16; mkdir -p $OUT
export TMPDIR=/tmp
for w in c5 paraffin_lamp wine_glass_1080p; do
  ACN_LIBDIR=$PWD/lib_phase timeout -k 10 300 python scripts/phase_ticks.py $w > $OUT/phase_ticks_$w.txt 2>&1; tail -n 45 $OUT/phase_ticks_$w.txt
done
echo session done
