#!/bin/bash
for w in 3 5 6; do
  make hip -B WALK_WAVES=$w > gpurun_out/build_sweep.log 2>&1 || { echo "build failed"; continue; }
  scripts/quick_bench.sh "walk_waves=$w"
done
make hip -B WALK_WAVES=4 SHADE_WAVES=6 > gpurun_out/build_sweep.log 2>&1 && scripts/quick_bench.sh "walk4 shade6"
make hip -B WALK_WAVES=4 SHADE_WAVES=8 > gpurun_out/build_sweep.log 2>&1 && scripts/quick_bench.sh "walk4 shade8"
