"""Debug helper (GPU box): find pixels where GPU and oracle differ and compare per-pixel work counters."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import actinon_amd as A
from oracle_binding import Oracle

def run(name, **ov):
    o = Oracle()
    sc = A.Scene.build(name, **ov)
    flat = sc.flatten()
    w, h = flat.params.image_width, flat.params.image_height
    pos = A.main_pass_positions(w, h)
    H = A.Handle(flat)
    gpu = H.render_positions(pos, linear=True)
    print(name, ov, "counters", H.last_counters())
    cpu = o.render_positions(flat, pos, linear=True)
    err = np.abs(gpu - cpu).max(axis=1)
    bad = np.nonzero(err > 1e-9)[0]
    print("bad pixels", len(bad), "of", len(pos), "max", err.max())
    for i in bad[:6]:
        g1 = H.render_positions(pos[i:i+1], linear=True)
        gc = H.last_counters()
        c1, cc = o.render_positions(flat, pos[i:i+1], linear=True, counters=True, threads=1)
        print(" pixel", i, pos[i], "gpu", g1[0], "cpu", c1[0])
        print("   gpu cnt", gc)
        print("   cpu cnt", {k: cc[k] for k in ["trans_ray", "shadow_ray", "obj_hit", "lum", "cap_sample", "side", "sdf_eval"]})
    H.close()

run("wine_glass", image_width=64, image_height=36, path_samples=16, direct_samples=50)
run("diamond", image_width=64, image_height=36, path_samples=32, direct_samples=50)
