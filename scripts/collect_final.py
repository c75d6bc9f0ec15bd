"""Copy the outputs of scripts/final_measure.sh (gpurun_out/<dir>) into profiles/: bench line, kernel-trace summaries,
traffic profile (stamped with the kernel source hash), per-kernel traffic table, logs.  usage: python scripts/collect_final.py <dir>"""
import csv, json, os, re, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
D = os.path.join(ROOT, "gpurun_out", sys.argv[1]); P = os.path.join(ROOT, "profiles")
t = json.load(open(os.path.join(D, "traffic_wine_glass_1080p.json")))
if t["kernel_source_hash"] != bench.kernel_source_hash():
    print("WARNING: traffic profile is of other kernel sources:", t["kernel_source_hash"], bench.kernel_source_hash())
old = json.load(open(os.path.join(P, "traffic.json")))
json.dump({"wine_glass_1080p": t, "_note": old["_note"]}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
tot = collections.defaultdict(lambda: [0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "pmc_" + c, "t_counter_collection.csv"))):
        if r["Counter_Name"] != c:
            continue
        m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        tot[m.group(2) + (m.group(3) or "")][i] += float(r["Counter_Value"])
lines, prod = [], 0
for k, (f, w) in sorted(tot.items(), key=lambda x: -(x[1][0] * 2 + x[1][1])):
    inst = bool(re.match(r"k_shade<\d+, true", k)) or bool(re.match(r"k_(walk|hard_path|hard_shadow|shade_hits)<true", k))
    passes = 1 if inst else 4
    gb = (f * 2 + w) * 1024 / passes / 1e9
    prod += 0 if inst else gb
    lines.append("%-40s fetch(x2) %7.1f GB  write %7.1f GB  total %7.1f GB per frame (%s kernel, %d pass%s of the run)" % (
        k, f * 2 * 1024 / passes / 1e9, w * 1024 / passes / 1e9, gb, "instrumented" if inst else "production", passes, "" if passes == 1 else "es"))
open(os.path.join(P, "r02", "traffic_by_kernel_final.txt"), "w").write(
    "HBM traffic per kernel and frame (PMC FETCH_SIZE x 2 + WRITE_SIZE, scripts/pmc_traffic.sh run of the closing batch;\n"
    "production kernels ran in 4 of the run's 5 passes, instrumented ones in 1).  Production kernels of one frame: %.1f GB\n" % prod + "\n".join(lines) + "\n")
for src, dst in (("bench_wine_glass_1080p.json", "bench_wine_glass_1080p.json"), ("stats4/s_kernel_stats.csv", "wine_glass_1080p_4lanes_kernel_stats.csv"),
                 ("stats1/s_kernel_stats.csv", "wine_glass_1080p_1lane_kernel_stats.csv"), ("tests_gpu.log", "tests_gpu.log"), ("smoke.log", "smoke.log")):
    shutil.copy(os.path.join(D, src), os.path.join(P, "r02", dst))
open(os.path.join(P, "r02", "bench_2ranks_one_gpu_rehearsal.json"), "w").write("".join(l for l in open(os.path.join(D, "bench_2ranks_rehearsal.json")) if l.startswith("{")))
d = json.loads([l for l in open(os.path.join(D, "bench_wine_glass_1080p.json")) if l.startswith("{")][-1])
print("production kernels %.1f GB/frame; run average %.1f GB" % (prod, t["hbm_bytes_per_step"] / 1e9))
print("value %.1f Msamples/s  %.2f ms/step  hbm frac %.4f (%.1f GB/s)  fp64 %.2f TFLOP/s  cpu %.2f Msamples/s  ratio %.1f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline_fp64"]["achieved"], d["cpu_baseline"]["value"], d["value"] / d["cpu_baseline"]["value"]))
for f in ("wine_glass_1080p_1lane_kernel_stats.csv",):
    for r in list(csv.DictReader(open(os.path.join(P, "r02", f))))[:8]:
        print("  %-45s calls %5s total %8.1f ms avg %8.3f ms" % (r["Name"][:45], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
