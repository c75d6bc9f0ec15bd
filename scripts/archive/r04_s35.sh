#!/bin/bash
# Closing measurements of round 4 on the re-built tree (the container was re-created; every .so is a fresh build of HEAD):
# GPU tests, smoke, kernel-trace summaries (default lanes / one lane), HBM traffic (two PMC passes, stamps profiles/traffic.json's
# copy under gpurun_out with the kernel source hash), the headline line, the 1/8 share, the frame under an 8 GiB workspace bound,
# c2, and the 2-rank rehearsals of both splits.        usage: scripts/r04_s35.sh <outdir under gpurun_out/>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1 || { tail -n 20 $out/tests_gpu.log; exit 1; }
tail -n 2 $out/tests_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -n 5 $out/smoke.log; exit 1; }
tail -n 2 $out/smoke.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats6 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $out/stats6.log 2>&1 || exit 1
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $out/stats1.log 2>&1 || exit 1
find $out/stats6 $out/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo traces done
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$c; mkdir -p $d
  rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 0 --quick --no-cpu-baseline > $d/log.txt 2>&1 || exit 1
done
python3 - "$out" <<'PY'
import csv, sys, json, os, re, collections, glob
out = sys.argv[1]
passes = 4     # --steps 4 --warmup 0 --quick: four production passes, nothing else
tot = {}; per = collections.defaultdict(lambda: [0.0, 0.0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and r["Kernel_Name"].startswith(("void k_", "k_"))]
    tot[c] = sum(float(r["Counter_Value"]) for r in rows)
    for r in rows:
        m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        per[m.group(2) + (m.group(3) or "")][i] += float(r["Counter_Value"])
fetch_b = tot["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads
write_b = tot["WRITE_SIZE"] * 1024
sys.path.insert(0, os.getcwd())
import bench
t = {"workload": "wine_glass_1080p", "passes": passes, "kernel_source_hash": bench.kernel_source_hash(), "FETCH_SIZE_KB_raw": tot["FETCH_SIZE"], "WRITE_SIZE_KB_raw": tot["WRITE_SIZE"],
     "hbm_bytes_per_step": (fetch_b + write_b) / passes, "fetch_bytes_per_step_x2": fetch_b / passes, "write_bytes_per_step": write_b / passes}
json.dump(t, open(os.path.join(out, "traffic_wine_glass_1080p.json"), "w"))
tj = json.load(open("profiles/traffic.json")); tj["wine_glass_1080p"] = t; json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
lines = ["HBM traffic per kernel and frame (PMC FETCH_SIZE x 2 + WRITE_SIZE, two separate rocprofv3 --pmc passes of", "bench.py --steps 4 --warmup 0 --quick: four production passes).  All kernels of one frame: %.1f GB" % ((fetch_b + write_b) / passes / 1e9)]
for k, (f, w) in sorted(per.items(), key=lambda x: -(x[1][0] * 2 + x[1][1])):
    lines.append("%-40s fetch(x2) %7.2f GB  write %7.2f GB  total %7.2f GB per frame" % (k, f * 2 * 1024 / passes / 1e9, w * 1024 / passes / 1e9, (f * 2 + w) * 1024 / passes / 1e9))
open(os.path.join(out, "traffic_by_kernel.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:7]))
PY
find $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE -name "*.csv" -delete
# the headline line: after the traffic passes, so that it carries the traffic of these very kernels
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --checksum $out/checksum_wine_glass_1080p.json > $out/bench_wine_glass_1080p.json 2> $out/bench.err || { tail -n 5 $out/bench.err; exit 1; }
cut -c1-200 $out/bench_wine_glass_1080p.json
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --quick --no-cpu-baseline --pixel-stride 8 > $out/bench_wine_glass_1080p_stride8.json 2>/dev/null || exit 1
cut -c1-160 $out/bench_wine_glass_1080p_stride8.json
for mb in 8192 16384; do
  ACN_WORKSPACE_MB=$mb timeout -k 10 300 python bench.py --steps 8 --warmup 3 --quick --no-cpu-baseline > $out/bench_wine_glass_1080p_ws$mb.json 2>/dev/null || exit 1
  cut -c1-160 $out/bench_wine_glass_1080p_ws$mb.json
done
timeout -k 10 300 python bench.py --workload c2 --steps 8 --warmup 2 --quick > $out/bench_c2.json 2> $out/bench_c2.err || { tail -n 5 $out/bench_c2.err; exit 1; }
cut -c1-160 $out/bench_c2.json
for split in tiles samples; do
  ACN_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline --split $split > $out/bench_2ranks_rehearsal_$split.json 2> $out/bench_2ranks_$split.err || { tail -n 5 $out/bench_2ranks_$split.err; exit 1; }
  grep '^{' $out/bench_2ranks_rehearsal_$split.json | cut -c1-200
done
echo final done
