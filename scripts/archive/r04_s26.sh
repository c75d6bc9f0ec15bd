#!/bin/bash
# round 4, GPU session 26: the 2-rank rehearsals again with three warm-up steps (the one re-allocation of a handle's queues happens in
# its second call and had fallen into the three timed steps of the closing session's rehearsal)
out=gpurun_out/s26; mkdir -p $out
export TMPDIR=/tmp
for split in tiles samples; do
  ACN_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline --split $split > $out/bench_2ranks_rehearsal_$split.json 2> $out/bench_2ranks_$split.err || { tail -n 5 $out/bench_2ranks_$split.err; exit 1; }
  grep '^{' $out/bench_2ranks_rehearsal_$split.json | cut -c1-200
done
