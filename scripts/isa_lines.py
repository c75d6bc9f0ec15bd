#!/usr/bin/env python3
"""Instruction counts of an AMDGPU assembly listing (-gline-tables-only) by source line, summed over buckets of lines:
usage: scripts/isa_lines.py kernel.s file.h:first-last[=name] ...   (prints instructions attributed to each range)"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
files = {}; loc = ("?", 0); hist = collections.Counter()
for ln in src:
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m: loc = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    if re.match(r"^(\.LBB\d+_\d+):", s): continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"): continue
    hist[loc] += 1
tot = sum(hist.values())
print("total", tot)
for spec in sys.argv[2:]:
    name = spec
    if "=" in spec: spec, name = spec.split("=")
    f, r = spec.split(":"); a, b = (int(v) for v in r.split("-"))
    n = sum(v for (ff, l), v in hist.items() if ff == f and a <= l <= b)
    print("%-40s %8d  %5.1f %%" % (name, n, 100.0 * n / tot))
