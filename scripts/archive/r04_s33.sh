#!/bin/bash
# round 4, GPU session 33: kernel-trace summaries of the closing build on many_spheres / 16 and diamond / 16 (one lane, one warm frame),
# and the issue counters of many_spheres again (after table, culling, second order, reservations)
set -o pipefail
OUT=$PWD/gpurun_out/s33; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
for w in c3 c4; do
  d=$OUT/trace_$w; mkdir -p $d
  ACN_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --workload $w --pixel-stride 16 --quick --steps 1 --warmup 1 --no-cpu-baseline > $d/log.txt 2>&1
  cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/${w}_stride16_1lane_kernel_stats_two_frames.csv
  find $d -name "*.csv" -size +5M -delete
done
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
  i=$((i+1)); d=$OUT/pmc_c3_$i; mkdir -p $d
  ACN_LANES=1 timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py --workload c3 --pixel-stride 16 --quick --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $OUT/pmc_c3_$i.txt
  find $d -name "*.csv" -size +5M -delete
done
python3 scripts/valu_table.py $OUT c3
echo session done
