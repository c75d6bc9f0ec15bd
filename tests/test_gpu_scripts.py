"""GPU side of the script front-end: scenes assembled by the .acn interpreter render
on the GPU to the oracle's result; the render driver behind scene.create_image writes the reference's PNM; the
actinon_hip command line tool stops softly on SIGINT and resumes from its recovery file."""
import os
import signal
import subprocess
import time

import numpy as np
import pytest

import actinon_amd as A
import scenes_util as S

pytestmark = pytest.mark.gpu
TOL = 1e-9
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCRIPTS = os.path.join(HERE, "scripts")
CLI = os.path.join(ROOT, "actinon_amd", "bin", "actinon_hip")


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if A.device_count() < 1:
        pytest.fail("no HIP device: the gpu tests need one")


def read_pnm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        data = np.frombuffer(f.read(), dtype=np.uint8)
    return data.reshape(h, w, 3)


@pytest.mark.parametrize("script,auto_env", [("csg.acn", A.Scene.AUTOENV_SKIP), ("nested.acn", A.Scene.AUTOENV_GPU),
                                             ("csg_light.acn", A.Scene.AUTOENV_SKIP)])
def test_script_scene_parity_with_oracle(oracle, script, auto_env):
    sc = A.Scene.from_script(os.path.join(SCRIPTS, script), auto_env)
    flat = sc.flatten()
    if script == "nested.acn":   # the estimator ran (on the GPU): the compound kept its two levels of envelopes
        top = [flat.node(i) for i in flat.elems_of(flat.c.matter_root)]
        assert [n.type for n in top].count(A.abi.ACN_COMPOUND) == 1
        assert all(flat.node(i).flags & A.abi.ACN_NODE_HAS_ENVELOPE for i in range(flat.n_nodes)
                   if flat.node(i).type == A.abi.ACN_COMPOUND and i not in (flat.c.matter_root, flat.c.light_root))
    pos = S.positions(flat)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    h.close()
    cpu = oracle.render_positions(flat, pos, linear=True)
    assert np.abs(gpu - cpu).max() <= TOL


# flattened scenes of the reference's scripts (tests/golden/scenes/*.npz), small frames: CSG-heavy node graphs that
# do not fit LDS staging (2847 / 4005 nodes), two lights (ruby_heart), deep glass stacks (pyramid)
FIXTURES = {
    "pyramid": dict(image_width=64, image_height=64, direct_samples=16, path_samples=16),
    "ruby_heart": dict(image_width=96, image_height=64, direct_samples=16, path_samples=8),
    "caustic_of_caustic": dict(image_width=64, image_height=64, direct_samples=16, path_samples=16),
    "paraffin_lamp": dict(image_width=48, image_height=72, direct_samples=8, path_samples=8),
    "hanging_lamp": dict(image_width=60, image_height=80, direct_samples=8, path_samples=8),
    "paraffin_lamp_on_ledge": dict(image_width=48, image_height=64, direct_samples=8, path_samples=8),
    "hanging_lamps_in_row": dict(image_width=64, image_height=36, direct_samples=4, path_samples=4),
}


@pytest.mark.parametrize("name", list(FIXTURES))
def test_fixture_scene_parity_with_oracle(oracle, name):
    flat = A.Flat.load(os.path.join(HERE, "golden", "scenes", name + ".npz"), **FIXTURES[name])
    pos = S.positions(flat)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    h.close()
    cpu = oracle.render_positions(flat, pos, linear=True)
    err = np.abs(gpu - cpu)
    assert err.max() <= TOL, f"{name}: {(err > TOL).any(axis=1).sum()} of {len(pos)} pixels differ, max {err.max():.3e}"


def test_run_script_renders_and_writes_pnm(oracle, tmp_path):
    """create_image through the default driver: main pass only (gradient_cycles 0) must equal the 8-bit image of the
    oracle's main pass."""
    src = open(os.path.join(SCRIPTS, "nested.acn")).read()
    p = tmp_path / "nested.acn"
    p.write_text(src)
    A.run_script(p)
    img = read_pnm(str(p) + ".pnm")
    sc = A.Scene.from_script(p, A.Scene.AUTOENV_GPU)
    flat = sc.flatten()
    cpu = oracle.render_positions(flat, S.positions(flat), linear=False)
    assert np.array_equal(img, A.cps_from_cl(cpu).reshape(img.shape))


def test_cli_sigint_saves_and_recovers(tmp_path):
    out = tmp_path / "cycles.pnm"
    tmp = str(out) + ".tmp.lum_image"
    cmd = [CLI, os.path.join(SCRIPTS, "cycles.acn"), str(out), "-f"]
    log = open(tmp_path / "run1.log", "wb")
    p = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT)
    try:
        deadline = time.time() + 300
        while time.time() < deadline:       # wait until a few gradient passes are through
            time.sleep(0.05)
            if b"gradient pass   3" in open(tmp_path / "run1.log", "rb").read():
                break
            assert p.poll() is None, open(tmp_path / "run1.log").read()
        p.send_signal(signal.SIGINT)
        rc = p.wait(timeout=120)
    finally:
        if p.poll() is None:
            p.kill()
    text = open(tmp_path / "run1.log").read()
    assert rc != 0 and "SIGINT received" in text and os.path.exists(tmp), text
    interrupted_at = max(int(l.split("gradient pass")[1].split(":")[0]) for l in text.splitlines() if "gradient pass" in l)
    assert interrupted_at < 400
    partial = read_pnm(out)                 # the image written by the last completed pass is intact

    # resume: -r picks the recovery file up and continues at the interrupted cycle; cut the script short through
    # a second SIGINT-free run would take all 400 cycles, so check the resume point and stop again
    log2 = open(tmp_path / "run2.log", "wb")
    p = subprocess.Popen(cmd + ["-r"], stdout=log2, stderr=subprocess.STDOUT)
    try:
        deadline = time.time() + 300
        seen = b""
        while time.time() < deadline:
            time.sleep(0.05)
            seen = open(tmp_path / "run2.log", "rb").read()
            if seen.count(b"gradient pass") >= 2:
                break
            assert p.poll() is None, seen.decode()
        p.send_signal(signal.SIGINT)
        p.wait(timeout=120)
    finally:
        if p.poll() is None:
            p.kill()
    text2 = open(tmp_path / "run2.log").read()
    assert f"resuming at gradient cycle {interrupted_at}" in text2, text2
    assert "main image" not in text2
    first = min(int(l.split("gradient pass")[1].split(":")[0]) for l in text2.splitlines() if "gradient pass" in l)
    assert first == interrupted_at
    assert read_pnm(out).shape == partial.shape


@pytest.mark.parametrize("name", ["hanging_lamp", "paraffin_lamp"])
def test_prune_programs_change_work_not_results(name, monkeypatch):
    """Interval-prune programs skip objects a ray cannot touch: fewer deferred rays, identical image."""
    flat = A.Flat.load(os.path.join(HERE, "golden", "scenes", name + ".npz"), **FIXTURES[name])
    pos = S.positions(flat)
    out = {}
    for label, min_nodes in (("off", "1000000000"), ("on", "32")):
        monkeypatch.setenv("ACN_PRUNE_MIN", min_nodes)      # read by acn_scene_upload
        h = A.Handle(flat)
        out[label] = (h.render_positions(pos, linear=True), h.last_stages()["hard_rays"])
        h.close()
    # not bit-equal: a sample finished inside k_shade joins the wave's floating-point sum, a deferred one is added to the
    # pixel on its own in 2^-40 fixed point; both are within 1e-11 of each other and 1e-9 of the oracle
    assert np.abs(out["on"][0] - out["off"][0]).max() < 1e-11
    assert out["on"][1] < out["off"][1]
