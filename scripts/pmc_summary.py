"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel name, sum of each counter."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", name)
    short = (m.group(2) + (m.group(3) or "")) if m else name[:30]
    agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if not k.startswith("k_"):
        continue
    print(k, {c: "%.4g" % x for c, x in v.items()})
    d = v
    if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"]:
        print("    VALU lane utilisation %.1f%%" % (100 * d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"])))
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_FLAT"):
            if c in d:
                print("    %s / WAVE_CYCLES = %.3f" % (c, d[c] / d["SQ_WAVE_CYCLES"]))
