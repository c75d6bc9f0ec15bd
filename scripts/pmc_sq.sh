#!/bin/bash
# SQ occupancy / stall counters of one short bench run (separate PMC pass, no trace flags)
# usage: scripts/pmc_sq.sh <workload> <tag> <counters...>
w=$1; tag=$2; shift 2
export TMPDIR=/tmp
d=gpurun_out/pmc_$tag
mkdir -p $d
rocprofv3 --pmc "$@" --output-format csv -d $d -o t -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
python3 scripts/pmc_summary.py $d/t_counter_collection.csv | tee $d/summary.txt
