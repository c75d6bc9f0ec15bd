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
                                             ("csg_light.acn", A.Scene.AUTOENV_SKIP),
                                             # texture maps / distance functions made by beth_object() in the script
                                             ("textured.acn", A.Scene.AUTOENV_SKIP)])
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


def independent_gradient_driver(flat, render, cycles, samples, threshold):
    """A second evaluation of scene_s_create_image_file's loop (src/scene.c:1103-1159) in numpy / Python, independent of
    actinon_amd/host/acn_driver.c: main pass, then per cycle the neighbour-gradient detector on the running averages
    (lum_image_s_sqr_grad, :848-862), `samples` jittered positions per flagged pixel from the lcg00 stream seeded 21943294
    (:799, 1132-1133), and the weighted accumulation of lum_image_s_push (:804-813).  render( pos ) -> gamma-saturated
    colours.  Returns the 8-bit image after the last pass."""
    w, h = int(flat.params.image_width), int(flat.params.image_height)
    clr = np.zeros((h, w, 3))
    wgt = np.zeros((h, w))
    rval = 21943294
    mask64 = (1 << 64) - 1
    a_, c_ = 6364136223846793005, 1442695040888963407      # ACN_LCG00_A / _C (include/actinon_hip.h)
    for cycle in range(cycles + 1):
        if cycle == 0:
            pos = A.main_pass_positions(w, h)
        else:
            f = np.where(wgt > 0, 1.0 / np.where(wgt > 0, wgt, 1.0), 1.0)
            avg = clr * f[..., None]                     # lum_image_s_get_avg multiplies by the reciprocal
            g = np.zeros((h, w))
            for dx, dy in ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)):
                sh = np.zeros_like(avg)
                ok = np.zeros((h, w), dtype=bool)
                ys, yd = (slice(max(dy, 0), h + min(dy, 0)), slice(max(-dy, 0), h + min(-dy, 0)))
                xs, xd = (slice(max(dx, 0), w + min(dx, 0)), slice(max(-dx, 0), w + min(-dx, 0)))
                sh[yd, xd] = avg[ys, xs]
                ok[yd, xd] = True
                d = avg - sh
                dev = (d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1]) + (d[..., 2] * d[..., 2])
                g = np.maximum(g, np.where(ok, dev, 0.0))
            flagged = np.argwhere(g > threshold * threshold)        # raster order: j outer, i inner
            pts = []
            for j, i in flagged:
                for _ in range(samples):
                    rval = (rval * a_ + c_) & mask64
                    dx = float(np.float64(np.uint64(rval))) * 2.0 ** -64
                    rval = (rval * a_ + c_) & mask64
                    dy = float(np.float64(np.uint64(rval))) * 2.0 ** -64
                    pts.append((i + dx, j + dy))
            pos = np.array(pts, dtype=np.float64).reshape(-1, 2)
        rgb = render(pos)
        for (px, py), c in zip(pos, rgb):
            x, y = int(px), int(py)                     # s2_t x = lum.pos.x / lum.weight, weight 1
            if 0 <= x < w and 0 <= y < h:
                clr[y, x] += c
                wgt[y, x] += 1.0
    f = np.where(wgt > 0, 1.0 / np.where(wgt > 0, wgt, 1.0), 1.0)
    return A.cps_from_cl(clr * f[..., None])


def test_gradient_cycles_match_an_independent_evaluation(oracle, tmp_path):
    """f-1: the driver's adaptive anti-aliasing against a second implementation fed with the ORACLE's colours."""
    script = tmp_path / "cycles3.acn"
    script.write_text(open(os.path.join(SCRIPTS, "cycles3.acn")).read())
    A.run_script(script)
    img = read_pnm(str(script) + ".pnm")
    sc = A.Scene.from_script(script, A.Scene.AUTOENV_GPU)
    flat = sc.flatten()
    assert sc.s.gradient_cycles == 3 and sc.s.gradient_samples == 3
    calls = []

    def render(pos):
        calls.append(len(pos))
        return oracle.render_positions(flat, pos, linear=False)

    ref = independent_gradient_driver(flat, render, 3, 3, sc.s.gradient_threshold)
    assert len(calls) == 4 and all(n > 0 for n in calls[1:]), calls      # the detector did flag pixels in every cycle
    assert calls[1] < calls[0] * 3                                       # ... and not all of them
    assert np.array_equal(img, ref.reshape(img.shape))
    # the passes changed the picture: it is not the main pass alone
    main = A.cps_from_cl(oracle.render_positions(flat, S.positions(flat))).reshape(img.shape)
    assert (img != main).any()


def run_cli_until(cmd, log_path, marker, deadline_s=300):
    """Starts the command line tool, sends SIGINT once `marker` shows up in its output; returns (rc, text)."""
    log = open(log_path, "wb")
    p = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT)
    try:
        deadline = time.time() + deadline_s
        while time.time() < deadline:
            time.sleep(0.02)
            if marker is not None and marker in open(log_path, "rb").read():
                p.send_signal(signal.SIGINT)
                break
            if p.poll() is not None:
                break
        rc = p.wait(timeout=deadline_s)
    finally:
        if p.poll() is None:
            p.kill()
    return rc, open(log_path).read()


def test_interrupted_and_resumed_render_equals_uninterrupted(tmp_path):
    """f-4: SIGINT in the middle of the gradient cycles, then `-r`: the final image is byte-identical to the one of a run
    that was never interrupted (the recovery file holds the accumulator, the cycle to redo and the jitter generator's
    state at its start; src/scene.c:1102-1106, 1143-1151)."""
    script = os.path.join(SCRIPTS, "cycles24.acn")
    full = tmp_path / "full.pnm"
    rc, text = run_cli_until([CLI, script, str(full), "-f"], tmp_path / "full.log", None)
    assert rc == 0 and "gradient pass  24" in text, text[-2000:]
    part = tmp_path / "part.pnm"
    rc, text = run_cli_until([CLI, script, str(part), "-f"], tmp_path / "part1.log", b"gradient pass   6")
    assert rc != 0 and "SIGINT received" in text and os.path.exists(str(part) + ".tmp.lum_image"), text[-2000:]
    stopped_at = max(int(l.split("gradient pass")[1].split(":")[0]) for l in text.splitlines() if "gradient pass" in l)
    assert 6 <= stopped_at < 24
    assert not np.array_equal(read_pnm(part), read_pnm(full))           # interrupted: not the final image yet
    rc, text2 = run_cli_until([CLI, script, str(part), "-f", "-r"], tmp_path / "part2.log", None)
    assert rc == 0 and f"resuming at gradient cycle {stopped_at}" in text2 and "gradient pass  24" in text2, text2[-2000:]
    assert np.array_equal(read_pnm(part), read_pnm(full))


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
