#!/usr/bin/env python3
"""Per kernel: share of the SIMDs' cycles in which a VALU instruction is active, active lanes per VALU instruction, share of the
waves' time spent waiting on a counter, resident waves per SIMD -- from two rocprofv3 --pmc passes summarised by pmc_summary.py
(SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_* / SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU), one lane (ACN_LANES=1).
SQ_BUSY_CYCLES is counted per shader engine (32), the wave / instruction counters in units of four cycles over 1024 SIMDs.
usage: scripts/valu_table.py <dir with pmc_<workload>_1.txt, pmc_<workload>_2.txt> workload [workload ...]"""
import re, ast, sys
d0 = sys.argv[1]
for wl in sys.argv[2:]:
    d = {}
    for i in (1, 2):
        try: txt = open("%s/pmc_%s_%d.txt" % (d0, wl, i)).read()
        except OSError: continue
        for m in re.finditer(r"^(\S.*?) (\{.*\})$", txt, re.M):
            d.setdefault(m.group(1), {}).update({k: float(v) for k, v in ast.literal_eval(m.group(2)).items()})
    print("== %s" % wl)
    print("%-38s %9s %6s %6s %6s %6s %10s %10s %10s %10s %10s" % ("kernel", "busy ms", "VALU%", "lanes%", "wait%", "waves", "VALU", "SALU", "VMEM_RD", "SMEM", "LDS"))
    for k, v in d.items():
        if v.get("SQ_BUSY_CYCLES", 0) < 1e8: continue
        quads = v["SQ_BUSY_CYCLES"] / 32 * 1024 / 4
        print("%-38s %9.1f %6.0f %6.0f %6.0f %6.1f %10.2e %10.2e %10.2e %10.2e %10.2e" % (k[:38], v["SQ_BUSY_CYCLES"] / 32 / 2.4e6,
              100 * v.get("SQ_ACTIVE_INST_VALU", 0) / quads, 100 * v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_ACTIVE_INST_VALU", 1), 1) / 64,
              100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAVE_CYCLES"] / quads, v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"],
              v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_SMEM"], v["SQ_INSTS_LDS"]))
