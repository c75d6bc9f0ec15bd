#!/usr/bin/env python3
"""Folds the per-frame digest files scripts/regen_digests.sh wrote on the GPU box into tests/golden/frame_checksums.json.
usage: scripts/merge_digests.py <dir with *.json> [note for _about]"""
import glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpath = os.path.join(root, "tests", "golden", "frame_checksums.json")
golden = json.load(open(gpath))
n = 0
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.load(open(f))
    for k, v in d.items():
        if isinstance(v, dict) and "sha256" in v and "/stride" in k:
            changed = golden.get(k, {}).get("sha256") != v["sha256"]
            golden[k] = v; n += 1
            print(("changed  " if changed else "same     ") + k)
if len(sys.argv) > 2:
    golden["_about"] = golden.get("_about", "") + "  " + sys.argv[2]
json.dump(golden, open(gpath, "w"))
print(n, "entries merged")
