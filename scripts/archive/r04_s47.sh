#!/bin/bash
# The chain of one lane on the share one of 8 ranks gets: kernel trace of the 1/8 share on ONE lane, and of 1/48 of the frame (what each of
# six lanes of that rank renders) alone on the chip -- per-dispatch timeline of the last run (scripts/trace_step.py).   usage: scripts/r04_s47.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for s in 8 48; do
  ACN_LANES=1 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$s -o t -- python3 bench.py --steps 3 --warmup 3 --quick --no-cpu-baseline --pixel-stride $s > $out/trace_$s.log 2>&1 || { tail -n 5 $out/trace_$s.log; exit 1; }
  f=$(find $out/trace_$s -name "*kernel_trace.csv" | head -n 1)
  python3 scripts/trace_step.py $f > $out/chain_stride$s.txt
  grep '^{' $out/trace_$s.log | cut -c1-200
  tail -n 12 $out/chain_stride$s.txt
  rm -rf $out/trace_$s
done
