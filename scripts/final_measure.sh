#!/bin/bash
# The round's closing measurements on one MI355X box (run through gpurun): GPU tests, smoke, the headline bench line with
# CPU baseline, kernel-trace summaries with 4 lanes and with one, HBM traffic, and a 2-rank rehearsal of the N > 1 path.
# usage: scripts/final_measure.sh <outdir under gpurun_out/>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1 || { tail -n 20 $out/tests_gpu.log; exit 1; }
tail -n 2 $out/tests_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -n 5 $out/smoke.log; exit 1; }
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > $out/bench_wine_glass_1080p.json 2> $out/bench.err || { tail -n 5 $out/bench.err; exit 1; }
cut -c1-200 $out/bench_wine_glass_1080p.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $out/stats4.log 2>&1 || exit 1
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $out/stats1.log 2>&1 || exit 1
scripts/pmc_traffic.sh wine_glass_1080p 2 > $out/traffic.log 2>&1 || { tail -n 5 $out/traffic.log; exit 1; }
cp gpurun_out/traffic_wine_glass_1080p.json $out/
tail -n 1 $out/traffic.log | cut -c1-300
ACN_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_2ranks_rehearsal.json 2> $out/bench_2ranks.err || { tail -n 5 $out/bench_2ranks.err; exit 1; }
grep '^{' $out/bench_2ranks_rehearsal.json | cut -c1-200
