"""Copy the outputs of scripts/final_measure_r04.sh (gpurun_out/<dir>) into profiles/r04/ and profiles/traffic.json, and print the
measurement table of DESIGN.md section 7 / BASELINE.md section 3.   usage: python scripts/collect_final_r04.py <dir>"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
D = os.path.join(ROOT, "gpurun_out", sys.argv[1]); P = os.path.join(ROOT, "profiles", "r04")
os.makedirs(P, exist_ok=True)
def line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        return None
t = json.load(open(os.path.join(D, "traffic_wine_glass_1080p.json")))
if t["kernel_source_hash"] != bench.kernel_source_hash():
    print("WARNING: traffic profile is of other kernel sources:", t["kernel_source_hash"], bench.kernel_source_hash())
old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
json.dump({"wine_glass_1080p": t, "_note": old.get("_note", "")}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(D, "bench_*.json")) + glob.glob(os.path.join(D, "checksum_*.json")) + [os.path.join(D, x) for x in ("traffic_by_kernel.txt", "tests_gpu.log", "smoke.log", "frames.txt")]:
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, os.path.basename(f)))
for src, dst in (("stats4", "wine_glass_1080p_6lanes_kernel_stats.csv"), ("stats1", "wine_glass_1080p_1lane_kernel_stats.csv")):
    f = glob.glob(os.path.join(D, src, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, dst))
d = line(os.path.join(D, "bench_wine_glass_1080p.json"))
print("headline %.1f Msamples/s  %.2f ms  cpu %s  speedup %.1f  hbm frac %.2e  fp64 %.2f TFLOP/s (%.3f)  traffic %.1f GB  ws %.1f GB  digest %s" % (
    d["value"], d["ms_per_step"], [round(b["value"], 2) for b in d["cpu_baseline"]["builds"]], d["speedup_vs_cpu_baseline"], d["roofline"]["frac"],
    d["roofline_fp64"]["achieved"], d["roofline_fp64"]["frac"], t["hbm_bytes_per_step"] / 1e9, d["stages"]["workspace_bytes"] / 1e9, d["frame_check"]["golden"]))
print("one-lane families:", {k: round(v["ms_per_pass"], 2) for k, v in d["roofline"]["kernel_families_one_lane"].items() if isinstance(v, dict)})
for w in ("c2", "c1", "c5", "paraffin_lamp", "wine_glass_1080p_stride8", "c3", "c4"):
    x = line(os.path.join(D, "bench_%s.json" % w))
    if x:
        cb = x.get("cpu_baseline")
        print("%-26s %10.2f ms  %9.2f %s  chunks %d  %s" % (w, x["ms_per_step"], x["value"], x["unit"], x["stages"]["chunks"],
              ("cpu %.4f (%d cores) x%.1f  digest %s" % (cb["value"], cb["cores"], x["speedup_vs_cpu_baseline"], (x.get("frame_check") or {}).get("golden"))) if cb else ""))
for w in ("c3_stride16_counted", "c4_stride16_counted", "c5full_stride256_counted"):
    x = line(os.path.join(D, "bench_%s.json" % w))
    if x and x.get("roofline_fp64"):
        r = x["roofline_fp64"]
        print("%-26s fp64 %.2f TFLOP/s (%.3f of peak), %.3e flop + %.3e transcendentals for %d pixels, %.1f ms" % (w, r["achieved"], r["frac"], r["flop"], r["transcendentals"], x["config"]["pixels"], x["ms_per_step"]))
for f in ("wine_glass_1080p_1lane_kernel_stats.csv",):
    p = os.path.join(P, f)
    if os.path.exists(p):
        for r in list(csv.DictReader(open(p)))[:8]:
            print("  %-45s calls %5s total %8.1f ms avg %8.3f ms" % (r["Name"][:45], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
