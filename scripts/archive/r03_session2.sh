#!/bin/bash
# GPU session 2 of round 3: parity suite, node-preload variant A/B, k_shade phase ticks, SMEM / VMEM latency counters
set -o pipefail
OUT=gpurun_out/s2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
B="--workload wine_glass_1080p --steps 8 --warmup 2 --quick"
echo "== 1080p: baseline / preload" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab.txt "$B" "ACN_X=0;ACN_LIBDIR=$PWD/lib_pre;ACN_X=1;ACN_LIBDIR=$PWD/lib_pre ACN_X=1"
echo "== stride 8: baseline / preload" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_stride8.txt "$B --pixel-stride 8" "ACN_X=0;ACN_LIBDIR=$PWD/lib_pre;ACN_LANES=1;ACN_LIBDIR=$PWD/lib_pre ACN_LANES=1"
echo "== other scenes: baseline / preload" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 3 --warmup 1 --quick" "ACN_X=0;ACN_LIBDIR=$PWD/lib_pre"
scripts/ab.sh $OUT/ab_c4.txt "--workload c4 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_LIBDIR=$PWD/lib_pre"
scripts/ab.sh $OUT/ab_c3.txt "--workload c3 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_LIBDIR=$PWD/lib_pre"
echo "== phase ticks" | tee -a $OUT/progress.txt
[ -d lib_pt ] && ACN_WORKSPACE_MB=65536 ACN_LIBDIR=$PWD/lib_pt timeout -k 10 300 python scripts/phase_ticks.py wine_glass_1080p > $OUT/phase_ticks_wine_glass.txt 2>&1
tail -40 $OUT/phase_ticks_wine_glass.txt
echo "== pmc" | tee -a $OUT/progress.txt
export ACN_LANES=1
mkdir -p $OUT/pmc_a $OUT/pmc_b
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_a -o t -- python3 bench.py --workload wine_glass_1080p --steps 1 --warmup 0 --no-cpu-baseline --quick > $OUT/pmc_a/log.txt 2>&1
python3 scripts/pmc_summary.py $(find $OUT/pmc_a -name "*counter_collection.csv" | head -1) > $OUT/pmc_a_summary.txt 2>&1; cat $OUT/pmc_a_summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_b -o t -- python3 bench.py --workload wine_glass_1080p --steps 1 --warmup 0 --no-cpu-baseline --quick > $OUT/pmc_b/log.txt 2>&1
python3 scripts/pmc_summary.py $(find $OUT/pmc_b -name "*counter_collection.csv" | head -1) > $OUT/pmc_b_summary.txt 2>&1; cat $OUT/pmc_b_summary.txt
echo done | tee -a $OUT/progress.txt
