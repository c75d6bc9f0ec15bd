"""Per-dispatch PMC table from rocprofv3 counter_collection.csv."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    name = r["Kernel_Name"]
    m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", name)
    short = (m.group(2) + (m.group(3) or "")) if m else name[:30]
    d = disp.setdefault(k, {"name": short, "grid": r.get("Grid_Size", ""), "vgpr": r.get("VGPR_Count", ""), "scratch": r.get("Scratch_Size", r.get("Private_Segment_Size", ""))})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in disp.items():
    if not d["name"].startswith("k_walk") and not d["name"].startswith("k_shade"):
        continue
    extra = ""
    if d.get("SQ_INSTS_VMEM_RD") and d.get("SQ_INST_LEVEL_VMEM"):
        extra += " vmem_lat %.0f" % (d["SQ_INST_LEVEL_VMEM"] / (d["SQ_INSTS_VMEM_RD"] + d.get("SQ_INSTS_VMEM_WR", 0)))
    if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES"):
        extra += " cyc/wave %.3g" % (d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"])
    print(k, d["name"], "grid", d["grid"], "vgpr", d["vgpr"], "scratch", d["scratch"], {c: "%.3g" % v for c, v in d.items() if c not in ("name", "grid", "vgpr", "scratch")}, extra)
