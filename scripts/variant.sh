#!/bin/bash
# Builds a variant of libactinon_hip.so for a same-box A/B: the objects of the main build, with the listed units recompiled
# from the CURRENT sources with extra flags.   usage: scripts/variant.sh <name> "<extra hipcc flags>" unit [unit ...]
# -> lib_<name>/libactinon_hip.so (+ a copy of libactinon_host.so); run with ACN_LIBDIR=$PWD/lib_<name>
set -e
name=$1; extra=$2; shift 2
mkdir -p build_$name lib_$name
cp -u build/*.o build_$name/
FLAGS="-DACN_SHADE_WAVES=4 -DACN_WALK_WAVES=4 -DACN_TRACE_WAVES=4 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -Iactinon_amd/csrc -std=c++17 -w"
pids=()
for u in "$@"; do
  sched=""; case $u in k_walk_lds|k_walk_glb) sched="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
  /opt/rocm/bin/hipcc $FLAGS $sched $extra -c -o build_$name/$u.o actinon_amd/csrc/$u.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_$name/libactinon_hip.so build_$name/*.o
cp actinon_amd/lib/libactinon_host.so lib_$name/
echo "lib_$name built"
