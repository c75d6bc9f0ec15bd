#!/bin/bash
# usage: scripts/final_heavy_r04.sh <outdir under gpurun_out/>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
# heavy configs at stated size: whole frames with digests and CPU sub-window baselines; work counters on every 16th pixel
for w in c3 c4; do
  timeout -k 10 400 python bench.py --workload $w --steps 1 --warmup 0 --quick --cpu-window 240x135 --checksum $out/checksum_$w.json > $out/bench_$w.json 2> $out/bench_$w.err || { tail -n 5 $out/bench_$w.err; exit 1; }
  cut -c1-160 $out/bench_$w.json
  timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --pixel-stride 16 > $out/bench_${w}_stride16_counted.json 2> $out/bench_${w}_s16.err || { tail -n 5 $out/bench_${w}_s16.err; exit 1; }
done
timeout -k 10 300 python bench.py --workload c5full --steps 1 --warmup 1 --no-cpu-baseline --pixel-stride 256 > $out/bench_c5full_stride256_counted.json 2> $out/bench_c5full_s256.err
echo heavy done
