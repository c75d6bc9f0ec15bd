"""Generates tests/golden/scenes/*.npz: flattened scenes (acn_flat_scene: nodes, elems, params) of the reference's
shipped scripts that have no hand-written builder in actinon_amd/host/acn_scenes.c.

Run in the container that has /root/reference (the GPU box has neither the scripts nor a way to get them):
    python tests/golden/make_scene_fixtures.py [name ...]
A fixture is the OUTPUT of our interpreter (actinon_amd/host/acn_interp.c) on the script -- geometry and parameters
as numbers -- not the script.  set_auto_envelope() calls inside the scripts are served by the oracle's estimator
(bit-identical to the GPU estimator, tests/test_gpu_parity.py::test_auto_envelope_matches_oracle) through the
acn_set_envelope_estimator test seam, because this container has no GPU.
"""
import ctypes as C
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import actinon_amd as A                      # noqa: E402
from actinon_amd._lib import host            # noqa: E402
from oracle_binding import Oracle            # noqa: E402

REF = "/root/reference/src_acn"
SCRIPTS = {
    "hanging_lamp": "hanging_lamp/hanging_lamp.acn",          # BASELINE.json configs[4]
    "paraffin_lamp": "paraffin_lamp/paraffin_lamp.acn",
    "paraffin_lamp_on_ledge": "paraffin_lamp_on_ledge/paraffin_lamp_on_ledge.acn",
    "pyramid": "pyramid.acn",
    "ruby_heart": "ruby_heart.acn",
    "caustic_of_caustic": "caustic_of_caustic.acn",
    "hanging_lamps_in_row": "hanging_lamps_in_row/hanging_lamps_in_row.acn",   # 28 439 nodes: lamps as nested compounds
    # frame 49 of the rotating-diamond video (angle = 25 + index): the reference ships image/diamond_video.acn.image_000049.png
    "diamond_video_049": "diamond_video.acn",
}

# Scripts that render a series of frames: which frame the fixture holds.  The script loops `def index = 0; while( index <
# 90 )` over create_image( index ); the interpreter's hook keeps the scene of the FIRST create_image call, so the loop
# bounds of a scratch copy of the script text (in a temporary directory, never committed) are set to the one frame.
FRAME_OF = {"diamond_video_049": ("def index = 0;", "def index = 49;", "while( index < 90 )", "while( index < 50 )")}


def script_path(name, rel, tmp):
    src = os.path.join(REF, rel)
    if name not in FRAME_OF:
        return src
    a0, a1, b0, b1 = FRAME_OF[name]
    text = open(src).read()
    assert text.count(a0) == 1 and text.count(b0) == 1, "loop of the frame series not found"
    dst = os.path.join(tmp, os.path.basename(rel))
    with open(dst, "w") as f:
        f.write(text.replace(a0, a1).replace(b0, b1))
    return dst


def main():
    oracle = Oracle()
    fn = C.cast(oracle.lib.acn_oracle_estimate_envelope, C.c_void_p)
    host.acn_set_envelope_estimator(fn)
    out_dir = os.path.join(HERE, "scenes")
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[1:]
    tmp = tempfile.mkdtemp(prefix="acn_fixture_")
    for name, rel in SCRIPTS.items():
        if only and name not in only:
            continue
        sc = A.Scene.from_script(script_path(name, rel, tmp), A.Scene.AUTOENV_GPU)
        flat = sc.flatten()
        path = os.path.join(out_dir, name + ".npz")
        flat.save(path, driver=(sc.s.gradient_threshold, sc.s.gradient_samples, sc.s.gradient_cycles))
        print(f"{name}: {flat.n_nodes} nodes, {flat.c.n_elems} elems, {sc.objects()} objects -> {os.path.getsize(path)} bytes")
    host.acn_set_envelope_estimator(None)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
