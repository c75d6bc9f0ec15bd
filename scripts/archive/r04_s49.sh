#!/bin/bash
# The private tail of a walk level on small shares: generations dealt to the whole chip again down to a smaller size (ACN_PRIVATE_LIMIT).
# usage: scripts/archive/r04_s49.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for v in 32768 8192 2048 512; do
  for s in 8 1; do
    echo -n "ACN_PRIVATE_LIMIT=$v --pixel-stride $s: " >> $out/ab_private_limit_shares.txt
    ACN_PRIVATE_LIMIT=$v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --quick --no-cpu-baseline --pixel-stride $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  walk launches %d' % (d['ms_per_step'], d['stages']['walk_launches']))" >> $out/ab_private_limit_shares.txt || exit 1
  done
done
done
cat $out/ab_private_limit_shares.txt
