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
    # texture fields (plain / chess; plane, sphere and distance projections), built in scenes_util.build_textured
    "textured": (None, None),
}


# the committed golden of these configs holds every GOLDEN_STRIDE-th pixel in x and y (keeps the fixture small)
GOLDEN_STRIDE = {"primitives_c1": 4}


def build_textured():
    """Texture fields (src/textures.c) on every object type that has a projection: chess floor plane, chess sphere
    (azimuth / elevation projection), plain-textured ellipsoid, a chess-textured light, chess on a torus (distance
    objects project to (0,0)).  No shipped .acn scene uses textures, so this scene exists only here."""
    import ctypes as C
    from actinon_amd._lib import host
    sc = A.Scene()
    sc.set(image_width=96, image_height=72, gamma=1.0, trace_depth=25, trace_min_intensity=0.03, direct_samples=20,
           path_samples=16, max_path_length=4.0, camera_position=(0, -8, 2), camera_view_direction=(0, 8, -2),
           camera_top_direction=(0, 0, 1), camera_focal_length=3, background_color=(0.3, 0.35, 0.4))
    objs = []
    light = host.acn_obj_sphere_s_create(0.6)
    host.acn_obj_set_radiance(light, 25.0)
    host.acn_obj_set_texture_field_chess(light, A.v3(1.0, 0.9, 0.8), A.v3(0.8, 0.9, 1.0), 3.0)
    host.acn_obj_move(light, A.v3(-2, -3, 5))
    objs.append(light)
    floor = host.acn_obj_plane_s_create()
    host.acn_obj_set_material(floor, b"diffuse_polished")
    host.acn_obj_set_texture_field_chess(floor, A.v3(0.9, 0.9, 0.9), A.v3(0.2, 0.2, 0.25), 1.0)
    host.acn_obj_move(floor, A.v3(0, 0, -1))
    objs.append(floor)
    ball = host.acn_obj_sphere_s_create(0.9)
    host.acn_obj_set_material(ball, b"diffuse")
    host.acn_obj_set_texture_field_chess(ball, A.v3(0.9, 0.2, 0.2), A.v3(0.95, 0.9, 0.3), 4.0)
    m = host.acn_rotx(25)
    host.acn_obj_rotate(ball, C.byref(m))
    host.acn_obj_move(ball, A.v3(-1.2, 0, 0))
    objs.append(ball)
    egg = host.acn_obj_squaroid_s_create_ellipsoid(0.5, 0.5, 0.8)
    host.acn_obj_set_material(egg, b"mirror")
    host.acn_obj_set_texture_field_plain(egg, A.v3(0.3, 0.8, 0.4))
    host.acn_obj_move(egg, A.v3(0.6, 0.5, -0.2))
    objs.append(egg)
    tor = host.acn_obj_torus_create(0.5, 0.18)
    host.acn_obj_set_material(tor, b"diffuse")
    host.acn_obj_set_texture_field_chess(tor, A.v3(0.2, 0.4, 0.9), A.v3(0.9, 0.9, 0.9), 2.0)
    host.acn_obj_move(tor, A.v3(1.7, -0.6, -0.5))
    objs.append(tor)
    for o in objs:
        sc.push(o)
        host.acn_obj_discard(o)
    return sc


def build(name):
    if name == "textured":
        sc = build_textured()
        return sc, sc.flatten()
    builder, ov = SMALL[name]
    sc = A.Scene.build(builder, **ov)
    return sc, sc.flatten()


def positions(flat):
    return A.main_pass_positions(flat.params.image_width, flat.params.image_height)


def block_means(img8, b=16):
    h, w, _ = img8.shape
    return img8[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).astype(np.float64).mean(axis=(1, 3))
