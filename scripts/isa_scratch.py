#!/usr/bin/env python3
"""Lists the scratch (spill) instructions of an AMDGPU assembly listing compiled with -gline-tables-only, each with the source line
(.loc) it belongs to, optionally restricted to the instruction range of one loop (as printed by isa_loops.py: start, len).
usage: scripts/isa_scratch.py kernel.s [start len]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = lo + int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 60
files = {}
loc = ("?", 0)
i = 0
for ln in src:
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m: loc = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    if re.match(r"^(\.LBB\d+_\d+):", s): continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"): continue
    if lo <= i < hi and s.startswith("scratch_"):
        print("%7d  %-22s %s" % (i, "%s:%d" % loc, s.split(";")[0].strip()))
    i += 1
