#!/bin/bash
# GPU session 20: closing numbers of the heavy configs; where the 1/8 share of the 1080p frame spends its 15 ms
set -o pipefail
OUT=$PWD/gpurun_out/s20
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2; do
  scripts/ab.sh $OUT/stride8_lanes.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "ACN_LANES=4;ACN_LANES=2;ACN_LANES=1;ACN_LANES=8"
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_stride8 -- python3 bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline --pixel-stride 8 > $OUT/trace_stride8.log 2>&1
python - <<'PY' > $OUT/stride8_timeline.txt 2>&1
import csv, glob, os
fs = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/s20/trace_stride8/**/*kernel_trace.csv"), recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "launches")
# the last frame: launches after the last k_camera_setup / first k_walk following a k_resolve
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
last = max(i for i, r in enumerate(t) if r[2].startswith("k_camera_setup") or "k_lane_gather" in r[2]) if any(r[2].startswith("k_camera_setup") or "k_lane_gather" in r[2] for r in t) else 0
# frames are separated by gaps > 0.3 ms with no kernel running
frames, cur, end = [], [], 0
for r in t:
    if cur and r[0] > end + 300000:
        frames.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
frames.append(cur)
print("frames by gaps:", [len(f) for f in frames])
f = frames[-1]
t0 = f[0][0]
busy = 0; e = t0
for r in f:
    s = max(r[0], e)
    if r[1] > s: busy += r[1] - s; e = r[1]
print("last frame: span %.2f ms, some kernel running %.2f ms, sum of kernel durations %.2f ms" % ((max(r[1] for r in f) - t0) / 1e6, busy / 1e6, sum(r[1] - r[0] for r in f) / 1e6))
byq = {}
for r in f: byq.setdefault(r[3], []).append(r)
for q, l in byq.items():
    gaps = sum(max(0, l[i + 1][0] - l[i][1]) for i in range(len(l) - 1))
    print("queue %s: %d launches, first %.2f ms, last end %.2f ms, kernel time %.2f ms, gaps between its launches %.2f ms" % (q, len(l), (l[0][0] - t0) / 1e6, (l[-1][1] - t0) / 1e6, sum(r[1] - r[0] for r in l) / 1e6, gaps / 1e6))
q0 = max(byq.values(), key=len)
for r in q0:
    print("  %8.3f .. %8.3f  %7.3f ms  %s" % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2]))
PY
head -12 $OUT/stride8_timeline.txt
bash scripts/final_heavy_r03.sh s20
echo done
