"""Aggregate a rocprofv3 kernel_trace.csv by short kernel name: calls, total ms, avg ms."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", name)
    short = (m.group(2) + (m.group(3) or "")) if m else name[:30]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1; a[1] += dur
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s calls %5d  total %9.3f ms  avg %8.3f ms  %5.1f%%" % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
