#!/bin/bash
# GPU session 3 of round 3: parity suite on the cone-culling kernels, headline bench line, heavy configs at stated size (c3, c4 full
# frames with frame digests and CPU sub-window baselines)
set -o pipefail
OUT=gpurun_out/s3
mkdir -p $OUT
export TMPDIR=/tmp
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
echo "== headline" | tee -a $OUT/progress.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --checksum $OUT/checksum_wine_glass_1080p.json > $OUT/bench_wine_glass_1080p.json 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/progress.txt
python - <<'PY' | tee -a $OUT/progress.txt
import json
d = json.load(open("gpurun_out/s3/bench_wine_glass_1080p.json"))
print("headline %.1f Msamples/s %.2f ms  x%.1f  check %s  ws %.2f GB" % (d["value"], d["ms_per_step"], d["speedup_vs_cpu_baseline"], d["frame_check"]["golden"], d["stages"]["workspace_bytes"] / 1e9))
print({k: round(v["ms_per_pass"], 2) for k, v in d["roofline"]["kernel_families_one_lane"].items() if isinstance(v, dict)})
PY
echo "== stride 8, c2, c1" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_small.txt "--workload wine_glass_1080p --steps 8 --warmup 2 --quick --pixel-stride 8" "ACN_X=0"
scripts/ab.sh $OUT/ab_small.txt "--workload c2 --steps 8 --warmup 2 --quick" "ACN_X=0"
scripts/ab.sh $OUT/ab_small.txt "--workload c1 --steps 20 --warmup 3 --quick" "ACN_X=0"
scripts/ab.sh $OUT/ab_small.txt "--workload c5 --steps 3 --warmup 1 --quick" "ACN_X=0"
scripts/ab.sh $OUT/ab_small.txt "--workload paraffin_lamp --steps 3 --warmup 1 --quick" "ACN_X=0"
echo "== c3 full frame" | tee -a $OUT/progress.txt
timeout -k 10 400 python bench.py --workload c3 --steps 1 --warmup 0 --quick --cpu-window 240x135 --checksum $OUT/checksum_c3.json > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "c3 rc $?" | tee -a $OUT/progress.txt
echo "== c4 full frame" | tee -a $OUT/progress.txt
timeout -k 10 400 python bench.py --workload c4 --steps 1 --warmup 0 --quick --cpu-window 240x135 --checksum $OUT/checksum_c4.json > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc $?" | tee -a $OUT/progress.txt
python - <<'PY' | tee -a $OUT/progress.txt
import json
for w in ("c3", "c4"):
    try:
        d = json.load(open("gpurun_out/s3/bench_%s.json" % w))
        print(w, "%.2f Msamples/s %.0f ms  cpu %.3f (%d cores) x%.1f  check %s" % (d["value"], d["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["speedup_vs_cpu_baseline"], d["frame_check"]["golden"]))
    except Exception as e:
        print(w, "failed", e)
PY
echo done | tee -a $OUT/progress.txt
