"""Shared helpers for the parity tests: small configurations of the BASELINE.json scenes."""
import numpy as np

import actinon_amd as A

# name -> (builder, overrides).  Sizes chosen so the CPU oracle finishes each in seconds.
SMALL = {
    # BASELINE configs[0]: primitives 400x300 path 0 direct 10 (full size: it is the CPU-runnable case)
    "primitives_c1": ("primitives", dict(image_width=400, image_height=300, path_samples=0, direct_samples=10)),
    # primitives with path tracing (torus SDF + squaroids under the path loop, max_path_length cut-off)
    "primitives_path": ("primitives", dict(image_width=96, image_height=72, path_samples=30, direct_samples=30)),
    # BASELINE configs[1] sampling (path 64 / direct 200) at reduced resolution
    "wine_glass_c2": ("wine_glass", dict(image_width=96, image_height=54, path_samples=64, direct_samples=200)),
    # diamond: deep refraction chains, 56-plane balanced CSG, chromatic reflection, tiny scene scale
    "diamond_c4": ("diamond", dict(image_width=64, image_height=36, path_samples=32, direct_samples=50)),
    # nested compounds with envelopes (3 levels = 512 spheres), CPU-only envelope substitute
    "many_spheres_c3": ("many_spheres:3:1", dict(image_width=64, image_height=36, path_samples=16, direct_samples=20)),
}


# the committed golden of these configs holds every GOLDEN_STRIDE-th pixel in x and y (keeps the fixture small)
GOLDEN_STRIDE = {"primitives_c1": 4}


def build(name):
    builder, ov = SMALL[name]
    sc = A.Scene.build(builder, **ov)
    return sc, sc.flatten()


def positions(flat):
    return A.main_pass_positions(flat.params.image_width, flat.params.image_height)


def block_means(img8, b=16):
    h, w, _ = img8.shape
    return img8[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).astype(np.float64).mean(axis=(1, 3))
