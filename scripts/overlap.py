"""Kernel overlap of a traced run: wall span vs sum of kernel durations, and busy time (union of intervals)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith(("void k_", "k_")))
# take the last third of the run (steady state)
t0, t1 = iv[0][0], iv[-1][1]
cut = t0 + (t1 - t0) * 2 // 3
iv = [x for x in iv if x[0] >= cut]
span = iv[-1][1] - iv[0][0]
total = sum(b - a for a, b in iv)
busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_b:
        busy += cur_b - cur_a; cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
print("kernels %d  span %.1f ms  busy(union) %.1f ms (%.0f%%)  sum of durations %.1f ms  (avg concurrency %.2f)" % (len(iv), span / 1e6, busy / 1e6, 100 * busy / span, total / 1e6, total / busy))
