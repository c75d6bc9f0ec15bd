"""Per-kernel HBM-side traffic from the two PMC passes of scripts/pmc_traffic.sh (FETCH_SIZE x2 per the gfx950
correction, WRITE_SIZE; KB -> GB), divided by the number of passes the bench run made."""
import csv, re, collections, sys
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tot = collections.defaultdict(lambda: [0.0, 0.0])
for c, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}/t_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        tot[m.group(2) + (m.group(3) or "")][idx] += float(r["Counter_Value"])
for k, (f, w) in sorted(tot.items(), key=lambda x: -x[1][0] - x[1][1])[:8]:
    print(f"{k:36s} fetch {f * 2048 / passes / 1e9:7.1f} GB  write {w * 1024 / passes / 1e9:7.1f} GB per pass")
