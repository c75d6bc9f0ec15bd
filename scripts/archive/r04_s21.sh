#!/bin/bash
# round 4, GPU session 21: many_spheres (C3) -- leaves of the table walk parked and evaluated wave-wide (DEFER): whole GPU suite,
# same-box A/B against the plain loop, the other workloads (must not move), VALU / wait counters of k_shade<64>
set -o pipefail
OUT=$PWD/gpurun_out/s21; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 8 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
python bench.py $W --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_sc_nodefer"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_c3.txt "$W" "$O;$M"
done
scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$O;$M;$O;$M"
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M"
echo "ab done" >> $OUT/progress.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS"; do
  i=$((i+1)); d=$OUT/pmc_$i; mkdir -p $d
  ACN_LANES=1 timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py $W --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $OUT/pmc_$i.txt
  find $d -name "*.csv" -size +5M -delete
  echo "pmc $i done" >> $OUT/progress.txt
done
grep -h "k_shade<64" $OUT/pmc_*.txt | cut -c1-400
echo session done
