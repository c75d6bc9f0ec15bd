#!/bin/bash
# round 4, GPU session 20: many_spheres (C3) -- the table walk requests its next entry before the leaf is evaluated, sphere leaves
# read from a compact table: parity, same-box A/B of the four combinations, and what the walk waits for (PMC, one lane)
set -o pipefail
OUT=$PWD/gpurun_out/s20; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "many_spheres or c3 or simple" > $OUT/tests_c3.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 6 $OUT/tests_c3.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
python bench.py $W --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_c3.txt "$W" "ACN_LIBDIR=$PWD/lib_sc_old;ACN_LIBDIR=$PWD/lib_sc_early;ACN_LIBDIR=$PWD/lib_sc_table;$M"
done
echo "ab done" >> $OUT/progress.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "FETCH_SIZE"; do
  i=$((i+1)); d=$OUT/pmc_$i; mkdir -p $d
  ACN_LANES=1 timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py $W --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $OUT/pmc_$i.txt
  find $d -name "*.csv" -size +5M -delete
  echo "pmc $i done" >> $OUT/progress.txt
done
grep -h "k_shade<64" $OUT/pmc_*.txt | cut -c1-400
echo session done
