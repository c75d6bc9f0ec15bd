#!/usr/bin/env python3
"""Oracle vs every shipped render of the reference (16x16 block means of tests/golden/ref_image_blocks.json), one line per
picture and variant -- the evidence table behind DESIGN.md section 5 / tests/test_reference_images.py.

    python scripts/ref_image_report.py [--full] > profiles/r03/ref_image_report.txt

Needs only the committed fixtures (no /root/reference).  --full renders every scene at its script's own sampling
(about 15 minutes on 8 cores); the default uses reduced sampling for the slow scenes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                 # noqa: E402
import actinon_amd as A            # noqa: E402
import scenes_util as S            # noqa: E402
from oracle_binding import Oracle  # noqa: E402

REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_image_blocks.json")))
FULL = "--full" in sys.argv


def fixture(name, **ov):
    return A.Flat.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"), **ov)


def no_sigma(nd):
    if nd.sigma > 0:
        nd.sigma = 0.0
        return 1
    return 0


def stand_default(nd):
    if nd.chromatic_reflectivity == 0.7:
        nd.diffuse_reflectivity = 1.0
        return 1
    return 0


def red(full, reduced):
    return full if FULL else reduced


# picture -> list of (variant, flat scene factory, node patch or None, block mask or None)
def gem_mask_diamond(shape):
    m = np.zeros(shape, dtype=bool); m[8:18, 2:24] = True; return m


def gem_mask_f49(shape):
    m = np.zeros(shape, dtype=bool); m[5:13, 7:18] = True; return m


CASES = [
    ("primitives", "as scripted", lambda: A.Scene.build("primitives").flatten(), None, None),
    ("primitives", "control: sigma = 0", lambda: A.Scene.build("primitives").flatten(), no_sigma, None),
    ("wine_glass", "as scripted" + red("", " (path 100)"), lambda: A.Scene.build("wine_glass", **red({}, dict(path_samples=100))).flatten(), None, None),
    ("wine_glass", "control: sigma = 0", lambda: A.Scene.build("wine_glass", path_samples=100).flatten(), no_sigma, None),
    ("many_spheres", red("as scripted 20 / 20", "12 / 8 samples"), lambda: A.Scene.build("many_spheres:5:1", **red({}, dict(path_samples=8, direct_samples=12))).flatten(), None, None),
    ("diamond", "as scripted", lambda: A.Scene.build("diamond").flatten(), None, gem_mask_diamond),
    ("diamond", "stand: diffuse_reflectivity 1.0 (objects.c:154)", lambda: A.Scene.build("diamond").flatten(), stand_default, gem_mask_diamond),
    ("diamond_video_049", "as scripted", lambda: fixture("diamond_video_049"), None, gem_mask_f49),
    ("pyramid", "as scripted", lambda: fixture("pyramid"), None, None),
    ("paraffin_lamp", red("as scripted", "20 / 20 samples"), lambda: fixture("paraffin_lamp", **red({}, dict(direct_samples=20, path_samples=20))), None, None),
    ("hanging_lamp", red("as scripted", "20 / 20 samples"), lambda: fixture("hanging_lamp", **red({}, dict(direct_samples=20, path_samples=20))), None, None),
    ("hanging_lamps_in_row", red("640 x 360, 30 / 30", "640 x 360, 10 / 8 samples"),
     lambda: fixture("hanging_lamps_in_row", image_width=640, image_height=360, **red({}, dict(direct_samples=10, path_samples=8))), None, None),
    ("paraffin_lamp_on_ledge", red("as scripted", "20 / 20 samples"), lambda: fixture("paraffin_lamp_on_ledge", **red({}, dict(direct_samples=20, path_samples=20))), None, None),
    ("ruby_heart", red("as scripted", "20 / 20 samples"), lambda: fixture("ruby_heart", **red({}, dict(direct_samples=20, path_samples=20))), None, None),
    ("ruby_heart", "sigma = 0", lambda: fixture("ruby_heart", **red({}, dict(direct_samples=20, path_samples=20))), no_sigma, None),
    ("caustic_of_caustic", red("as scripted", "20 / 20 samples"), lambda: fixture("caustic_of_caustic", **red({}, dict(direct_samples=20, path_samples=20))), None, None),
    ("caustic_of_caustic", "sigma = 0", lambda: fixture("caustic_of_caustic", **red({}, dict(direct_samples=20, path_samples=20))), no_sigma, None),
]


def main():
    o = Oracle()
    threads = os.cpu_count() or 1
    print(f"# oracle (oracle/libacn_oracle.so) vs the reference's shipped renders, 16 x 16 block means, 8-bit steps; {threads} threads")
    print("# picture | variant | W x H, direct / path | nodes patched | mean |d|  bias  max |d|  [ masked blocks: mean |d| bias max | rest: mean |d| bias max ] | s")
    for name, variant, make, patch, mask in CASES:
        flat = make()
        n = 0
        if patch:
            for i in range(flat.n_nodes):
                n += patch(flat.node(i))
        w, h = flat.params.image_width, flat.params.image_height
        t0 = time.time()
        img = o.render_positions(flat, A.main_pass_positions(w, h), threads=threads)
        dt = time.time() - t0
        d = S.block_means(A.cps_from_cl(img).reshape(h, w, 3), REF["block"]) - np.array(REF["images"][name]["block_means"])
        line = f"{name} | {variant} | {w} x {h}, {flat.params.direct_samples} / {flat.params.path_samples} | {n} | {np.abs(d).mean():.3f} {d.mean():+.3f} {np.abs(d).max():.2f}"
        if mask:
            m = mask(d.shape[:2])
            line += f" | masked {int(m.sum())}: {np.abs(d[m]).mean():.3f} {d[m].mean():+.3f} {np.abs(d[m]).max():.2f} | rest: {np.abs(d[~m]).mean():.3f} {d[~m].mean():+.3f} {np.abs(d[~m]).max():.2f}"
        print(line + f" | {dt:.1f}", flush=True)


if __name__ == "__main__":
    main()
