"""Summarise a rocprofv3 kernel_trace.csv: per-dispatch durations in order (short kernel names)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = []
for r in rows:
    name = r["Kernel_Name"]
    m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", name)
    short = (m.group(2) + (m.group(3) or "")) if m else name[:30]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out.append((short, dur, r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("Scratch_Size", r.get("Private_Segment_Size", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", "")))
for o in out[skip:]:
    if o[1] > 0.05:
        print("%-28s %9.3f ms  vgpr %s agpr %s sgpr %s scratch %s grid %s" % o[:7])
