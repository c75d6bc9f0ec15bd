#!/bin/bash
# Hypothesis: on an under-filled chip (the share one of 8 ranks gets) shading tasks on 64 lanes instead of 16 shorten the chain.
# usage: scripts/r04_s48.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for v in 255 100 32; do
  for s in 8 4; do
    echo -n "ACN_CLASS0_MIN=$v --pixel-stride $s: " >> $out/ab_class0_shares.txt
    ACN_CLASS0_MIN=$v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --quick --no-cpu-baseline --pixel-stride $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['ms_per_step'])" >> $out/ab_class0_shares.txt || exit 1
  done
done
done
cat $out/ab_class0_shares.txt
