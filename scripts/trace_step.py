"""Per-dispatch timeline of the LAST `k_finalize`-terminated pipeline run in a rocprofv3 kernel_trace.csv:
    python scripts/trace_step.py <kernel_trace.csv>
prints, in launch order, start offset / duration / gap to the previous dispatch of every pipeline kernel of that run."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(name):
    m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", name)
    return (m.group(2) + (m.group(3) or "")) if m else name[:30]
ends = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_finalize")]
if len(ends) < 2:
    print("need two k_finalize dispatches"); sys.exit(1)
seg = rows[ends[-2] + 1: ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot = {}
for r in seg:
    n = short(r["Kernel_Name"]); s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    d = (e - s) / 1e6
    tot[n] = tot.get(n, 0) + d
    if d > 0.02 or n.startswith("k_"):
        print("%9.3f ms  +%8.3f ms  gap %7.3f  %s" % ((s - t0) / 1e6, d, (s - prev_end) / 1e6, n))
    prev_end = max(prev_end, e)
print("span %.3f ms" % ((prev_end - t0) / 1e6))
for n, d in sorted(tot.items(), key=lambda x: -x[1]):
    print("  %-40s %9.3f ms" % (n, d))
