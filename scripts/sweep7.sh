#!/bin/bash
for d in "-DACN_TRACE_WAVES=3" "-DACN_TRACE_WAVES=3 -DACN_HPATH_WAVES=3" "-DACN_HPATH_WAVES=3" "-DACN_TRACE_WAVES=2"; do
  make hip -B EXTRA_DEFS="$d" > gpurun_out/build_sweep.log 2>&1 || { echo "build failed"; continue; }
  scripts/quick_bench.sh "$d"
done
