#!/usr/bin/env python3
"""Where a kernel's spill traffic sits: for every loop of an AMDGPU assembly listing (a backward branch to a label), its length in
instructions and the scratch loads / stores, SGPR-spill lane moves (v_writelane / v_readlane), global accesses and fp64 divisions
inside it.   usage: scripts/isa_loops.py kernel.s [min_instructions]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 20
labels = {}; ins = []
for ln in src:
    s = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m: labels[m.group(1)] = len(ins); continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"): continue
    ins.append(s)
loops = []
for i, s in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
    if m and m.group(1) in labels and labels[m.group(1)] <= i: loops.append((labels[m.group(1)], i, m.group(1)))
def cnt(a, b, pat): return sum(1 for s in ins[a:b + 1] if re.match(pat, s))
print("total instructions %d  scratch_load %d  scratch_store %d  writelane %d  readlane %d" % (len(ins), cnt(0, len(ins), "scratch_load"), cnt(0, len(ins), "scratch_store"), cnt(0, len(ins), "v_writelane"), cnt(0, len(ins), "v_readlane")))
print("%-14s %8s %8s %6s %6s %6s %6s %6s %6s %6s" % ("loop", "start", "len", "sld", "sst", "wlane", "rlane", "gload", "gstore", "div"))
for a, b, l in sorted(set(loops), key=lambda x: (x[0], -x[1])):
    if b - a < minlen: continue
    print("%-14s %8d %8d %6d %6d %6d %6d %6d %6d %6d" % (l, a, b - a + 1, cnt(a, b, "scratch_load"), cnt(a, b, "scratch_store"), cnt(a, b, "v_writelane"), cnt(a, b, "v_readlane"), cnt(a, b, "global_load"), cnt(a, b, "global_store"), cnt(a, b, "v_div_fmas_f64")))
