#!/bin/bash
# k_walk at 2 waves per SIMD (256 VGPRs, fewer scratch reloads per step) against the production 4 on SMALL shares, where a step's latency
# and not the chip's throughput bounds the chain.  variants_t2/lib: make LIBDIR=... TRACE_WAVES=2.   usage: scripts/archive/r04_s50.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for lib in actinon_amd/lib variants_t2/lib; do
  for cfg in "8 6" "4 6" "2 6" "1 6" "48 1"; do
    set -- $cfg
    echo -n "$lib --pixel-stride $1 ACN_LANES=$2: " >> $out/ab_trace_waves_shares.txt
    ACN_LIBDIR=$PWD/$lib ACN_LANES=$2 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --quick --no-cpu-baseline --pixel-stride $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['ms_per_step'])" >> $out/ab_trace_waves_shares.txt || exit 1
  done
done
done
cat $out/ab_trace_waves_shares.txt
