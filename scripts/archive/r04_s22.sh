#!/bin/bash
# round 4, GPU session 22: which kernels are short of VALU issue slots and how many lanes their VALU instructions carry
# (one lane, so that the kernels of a frame do not overlap): 1080p headline, diamond / 16, hanging_lamp 600x800
set -o pipefail
OUT=$PWD/gpurun_out/s22; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for wl in "1080p:--workload wine_glass_1080p" "c4:--workload c4 --pixel-stride 16" "c5:--workload c5" "c3:--workload c3 --pixel-stride 16"; do
  name=${wl%%:*}; W=${wl#*:}
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
    i=$((i+1)); d=$OUT/pmc_${name}_$i; mkdir -p $d
    ACN_LANES=1 timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py $W --quick --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 scripts/pmc_summary.py $f > $OUT/pmc_${name}_$i.txt
    find $d -name "*.csv" -size +5M -delete
    echo "pmc $name $i done" >> $OUT/progress.txt
  done
done
scripts/ab.sh $OUT/c3_table.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "ACN_LIBDIR=$PWD/actinon_amd/lib"
echo session done
