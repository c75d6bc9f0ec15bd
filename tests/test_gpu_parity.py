"""GPU parity (run on the MI355X box: pytest -m gpu).  Everything goes through the C ABI of libactinon_hip.so.

Bar: the HIP path must reproduce the CPU oracle's image.  Geometry, seeds, sample counts and control flow are
bit-identical by construction (same fp64 expressions, no contraction, shared deterministic transcendentals); only
the radiance summation is re-associated, so colours may move by a few ulp.  TOL is the stated per-channel tolerance:
1e-9 absolute on linear radiance and on the gamma-saturated output (BASELINE.json asks for <= 1e-3)."""
import ctypes as C
import os

import re
import numpy as np
import pytest

import actinon_amd as A
import scenes_util as S
from test_detmath import cpu_eval

pytestmark = pytest.mark.gpu

TOL = 1e-9
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_images.npz")


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert A.device_count() >= 1, "no HIP device: the gpu tests must run on the GPU box"


def test_detmath_bit_identical_cpu_gpu(detmath_cpu):
    """The arithmetic contract: every deterministic kernel, IEEE divide, sqrt and the u64->f64 conversion return
    the same bits on gfx950 as on the host."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    cases = {
        "sin": rng.uniform(-7, 7, n), "cos": rng.uniform(-7, 7, n), "tan": rng.uniform(0, np.pi, n),
        "acos": rng.uniform(-1, 1, n), "log": np.exp(rng.uniform(-30, 30, n)), "exp": rng.uniform(-700, 700, n),
        "sqrt": np.exp(rng.uniform(-300, 300, n)), "frexp_mant": rng.normal(size=n) * 10.0 ** rng.integers(-300, 300, n),
    }
    for op, x in cases.items():
        g = A.detmath_eval(op, x)
        c = cpu_eval(detmath_cpu, op, x)
        assert np.array_equal(g.view(np.uint64), c.view(np.uint64)), op
    x = np.exp(rng.uniform(-300, 300, n)) * rng.choice([-1, 1], n)
    y = np.exp(rng.uniform(-300, 300, n))
    assert np.array_equal(A.detmath_eval("div", x, y).view(np.uint64), (x / y).view(np.uint64))
    assert np.array_equal(A.detmath_eval("sqrt", y).view(np.uint64), np.sqrt(y).view(np.uint64))
    px, py = rng.uniform(0, 1, n), rng.uniform(0, 8, n)
    assert np.array_equal(A.detmath_eval("pow", px, py).view(np.uint64), cpu_eval(detmath_cpu, "pow", px, py).view(np.uint64))
    u = rng.integers(0, 2 ** 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)
    assert np.array_equal(A.detmath_eval("u64_to_f64", u.view(np.float64)), u.astype(np.float64))
    sub = np.array([5e-324, 2.2e-308, 1e-310, 0.0, -0.0])
    assert np.array_equal(A.detmath_eval("sqrt", np.abs(sub)).view(np.uint64), np.sqrt(np.abs(sub)).view(np.uint64))


@pytest.mark.parametrize("count_work", [False, True], ids=["production_kernels", "instrumented_kernels"])
@pytest.mark.parametrize("name", list(S.SMALL))
def test_image_parity_with_oracle(oracle, name, count_work):
    """Both builds of the pipeline kernels -- the ones every render uses and the ones ACN_OPT_COUNT_WORK selects --
    against the oracle, pixel by pixel."""
    sc, flat = S.build(name)
    pos = S.positions(flat)
    h = A.Handle(flat, count_work=count_work)
    for linear in (True, False):
        gpu = h.render_positions(pos, linear=linear)
        cpu = oracle.render_positions(flat, pos, linear=linear)
        err = np.abs(gpu - cpu)
        bad = (err > TOL).any(axis=1).sum()
        assert bad == 0, f"{name} linear={linear}: {bad} of {len(pos)} pixels differ, max {err.max():.3e}"
    # integer pixel indexing identical: the 8-bit image is the same
    assert np.array_equal(A.cps_from_cl(gpu), A.cps_from_cl(cpu))
    h.close()


@pytest.mark.parametrize("name", list(S.SMALL))
def test_image_parity_with_committed_golden(name):
    gold = np.load(GOLD)[name]
    sc, flat = S.build(name)
    pos = S.positions(flat)
    sub = S.GOLDEN_STRIDE.get(name, 1)
    if sub > 1:
        w, hh = flat.params.image_width, flat.params.image_height
        pos = pos.reshape(hh, w, 2)[::sub, ::sub].reshape(-1, 2)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True).reshape(gold.shape)
    h.close()
    assert np.abs(gpu - gold).max() <= TOL


def test_work_counters_match_oracle_exactly(oracle):
    """Control flow is bit-identical: the GPU casts exactly as many rays / samples as the oracle."""
    sc, flat = S.build("wine_glass_c2")
    pos = S.positions(flat)
    h = A.Handle(flat, count_work=True)
    h.render_positions(pos)
    g = h.last_counters()
    h.close()
    _, c = oracle.render_positions(flat, pos, counters=True)
    assert g["lum_calls"] == c["lum"]
    assert g["cap_samples"] == c["cap_sample"]
    assert g["shadow_rays"] == c["shadow_ray"]
    assert g["trans_rays"] == c["trans_ray"]
    # fp64 work by the cost table of SURVEY.md App. B (actinon_amd/csrc/acn_costs.h): the oracle tallies the reference's
    # algorithm, the device what it executed -- the same events, minus what any-hit shadow tests and envelope pruning
    # let it skip.  Per-sample shading work (2 transcendentals per cap sample, ...) is not skippable -- except the three
    # transcendentals of the Oren-Nayar weight in the DIRECT-light loop, whose closed form needs none
    # (oren_nayar_weight_direct in acn_device.h: that weight only scales a colour); the path loop keeps the exact form.
    assert 0.5 * c["flop"] < g["flop"] <= c["flop"], (g["flop"], c["flop"])
    assert 0.97 * (c["transc"] - 3 * c["oren_nayar"]) <= g["transcendentals"] < c["transc"], (g["transcendentals"], c["transc"], c["oren_nayar"])


def test_probe_rays_replace_walks_without_changing_counts(oracle):
    """Specular children that scene_s_lum would answer with zero (src/scene.c:430) are any-hit probes, not walks: a good
    part of the wine glass's rays, and the sum of both is what the pipeline walked when every ray was a walk -- the number
    of scene_s_trans_hit calls of the oracle stays exactly matched (test_work_counters_match_oracle_exactly)."""
    sc, flat = S.build("wine_glass_c2")
    pos = S.positions(flat)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    st = h.last_stages()
    h.close()
    assert st["probe_rays"] > 0.2 * st["walk_rays"], st
    cpu = oracle.render_positions(flat, pos, linear=True)
    assert np.abs(gpu - cpu).max() <= TOL


def test_edge_cases(oracle):
    # empty scene -> background everywhere; zero positions; sub-pixel / off-raster positions
    sc = A.Scene()
    sc.set(image_width=8, image_height=6, background_color=(0.1, 0.2, 0.3), camera_view_direction=(0, 1, 0),
           camera_top_direction=(0, 0, 1))
    flat = sc.flatten()
    h = A.Handle(flat)
    # (pixel sums are 2^-40 fixed point: 9.1e-13 resolution)
    assert np.abs(h.render_positions(S.positions(flat)) - np.tile([0.1, 0.2, 0.3], (48, 1))).max() <= 1e-12
    assert h.render_positions(np.zeros((0, 2))).shape == (0, 3)
    h.close()
    sc, flat = S.build("primitives_path")
    h = A.Handle(flat)
    pos = np.array([[-3.25, 100.5], [1e6, -1e6], [3.999, 2.001], [47.3, 35.7], [48.0, 36.0]])
    assert np.abs(h.render_positions(pos) - oracle.render_positions(flat, pos)).max() <= TOL
    h.close()


def test_main_pass_device_entry_point_and_subranges(oracle):
    """acn_render_main_pass_dev generates pixel centres on the device; any pixel sub-range gives the same pixels."""
    import torch
    sc, flat = S.build("wine_glass_c2")
    w, hh = flat.params.image_width, flat.params.image_height
    n = w * hh
    h = A.Handle(flat)
    ref = h.render_positions(S.positions(flat), linear=True)
    out = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
    h.render_main_pass_dev(0, n, out.data_ptr(), linear=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    out.zero_()
    first, count = 1234, 777
    h.render_main_pass_dev(first, count, out[first:].data_ptr(), linear=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[first:first + count], ref[first:first + count])
    with pytest.raises(A.AcnError):
        h.render_main_pass_dev(n - 5, 10, out.data_ptr())
    h.close()


def test_run_to_run_determinism_and_gamma_consistency():
    sc, flat = S.build("diamond_c4")
    pos = S.positions(flat)
    h = A.Handle(flat)
    a = h.render_positions(pos, linear=True)
    b = h.render_positions(pos, linear=True)
    assert np.array_equal(a, b)
    g = h.render_positions(pos, linear=False)
    gamma = flat.params.gamma
    assert np.allclose(g, np.clip(np.power(np.maximum(a, 0), gamma), 0, 1), atol=1e-12)
    h.close()


def test_full_size_properties():
    """BASELINE configs[1] at full size (wine_glass 1280x720, path 64 / direct 200): properties that do not need the
    oracle -- finite, in range, deterministic on a strided pixel subset, and independent of which other pixels are
    rendered in the same launch."""
    sc = A.Scene.build("wine_glass", image_width=1280, image_height=720, path_samples=64, direct_samples=200)
    flat = sc.flatten()
    h = A.Handle(flat)
    pos = A.main_pass_positions(1280, 720)
    full = h.render_positions(pos)
    assert np.isfinite(full).all() and full.min() >= 0 and full.max() <= 1
    idx = np.arange(0, len(pos), 97)
    sub = h.render_positions(pos[idx])
    assert np.array_equal(sub, full[idx])
    # the glass and its caustic are there: the frame is not flat
    assert full.reshape(720, 1280, 3)[:, :, 0].std() > 0.05
    h.close()


def test_auto_envelope_matches_oracle(oracle):
    """obj_estimate_envelope (objects.c:312-363) on the device == oracle, bit for bit (it feeds scene geometry)."""
    from actinon_amd._lib import host
    objs = [host.acn_obj_sphere_s_create(0.025), host.acn_obj_squaroid_s_create_ellipsoid(0.3, 0.3, 0.5)]
    host.acn_obj_move(objs[0], A.v3(0.3, -0.2, 0.1))
    for o in objs:
        flat = A.Flat()
        node = C.c_int32()
        A.check(host.acn_obj_flatten(o, C.byref(flat.c), C.byref(node)), "flatten")
        flat._owned = True
        h = A.Handle(flat)
        g = h.estimate_envelope(node.value)
        h.close()
        c = oracle.estimate_envelope(flat, node.value)
        assert g == c, (g, c)
        host.acn_obj_discard(o)


def test_many_spheres_with_gpu_auto_envelopes(oracle):
    """many_spheres.acn built the reference's way (Monte-Carlo auto envelopes run on the GPU), 3 levels."""
    sc = A.Scene.build("many_spheres:3:0", image_width=64, image_height=36, path_samples=16, direct_samples=20)
    flat = sc.flatten()
    pos = S.positions(flat)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    h.close()
    cpu = oracle.render_positions(flat, pos, linear=True)
    assert np.abs(gpu - cpu).max() <= TOL


@pytest.mark.parametrize("spec", ["many_spheres:3:0", "many_spheres:4:1"], ids=["monte_carlo_envelopes", "analytic_envelopes"])
def test_culled_table_walk_renders_the_same_bits(spec, monkeypatch, capfd):
    """simple_compound_hit skips an entry whose verified bounding envelope lies wholly behind the best hit so far (the
    reference visits it: compound.c:215-243), and walks a second table with every compound's children in reverse order when
    the ray runs against the order of the first ("first leaf wins a tie" becomes "last visited wins").  Neither can change
    the minimum or the leaf it belongs to, so the frame is the one of the plain walk to the bit -- and the upload step does
    find envelopes it can verify in both ways of building the scene."""
    sc = A.Scene.build(spec, image_width=96, image_height=54, path_samples=16, direct_samples=20)
    flat = sc.flatten()
    pos = S.positions(flat)
    frames = {}
    for label, env in (("culled_two_orders", dict(ACN_VERBOSE="1")), ("culled_one_order", dict(ACN_NO_SC_REVERSED="1", ACN_VERBOSE="1")),
                       ("plain", dict(ACN_NO_SC_CULL="1", ACN_VERBOSE="1"))):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = A.Handle(flat)
        frames[label] = h.render_positions(pos, linear=True)
        h.close()
        for k in env:
            monkeypatch.delenv(k)
        err = capfd.readouterr().err
        m = re.search(r"simple compounds: (\d+) entries, (\d+) with a verified bounding envelope, (\d+) again in reversed order", err)
        assert m, err
        entries, bounding, reversed_ = (int(m.group(k)) for k in (1, 2, 3))
        assert entries > 500
        assert (bounding > entries // 2) if label != "plain" else bounding == 0
        assert reversed_ == (entries if label == "culled_two_orders" else 0)
    assert np.array_equal(frames["culled_two_orders"], frames["plain"])
    assert np.array_equal(frames["culled_one_order"], frames["plain"])


def test_lanes_grid_and_walk_arrangement_do_not_change_a_pixel(oracle, monkeypatch):
    """Concurrent lanes (ACN_LANES), the size of the persistent grid (ACN_GRID), how the specular walk is cut into
    generation passes and private-stack finishing (ACN_WALK_PASSES, ACN_PRIVATE_LIMIT) and the overflow of the private
    stacks (forced by letting a pass use one slot of each wave's stack, ACN_TEST_STACK_USE) reorganise the work, not the
    arithmetic: the frame is bit-identical in every arrangement and equals the oracle on a sample of pixels."""
    sc = A.Scene.build("wine_glass", image_width=320, image_height=180, path_samples=16, direct_samples=50)
    flat = sc.flatten()
    pos = S.positions(flat)
    assert len(pos) >= 4 * 32 * 256          # enough tiles for four lanes
    frames = {}
    for label, env in (("plain", dict(ACN_LANES="1")), ("lanes", dict(ACN_LANES="4")),
                       ("small_grid", dict(ACN_LANES="1", ACN_GRID="7", ACN_SHADE_GRID="5")),
                       ("stack_overflow", dict(ACN_LANES="1", ACN_TEST_STACK_USE="1")),
                       ("all_private", dict(ACN_LANES="1", ACN_WALK_PASSES="1")),
                       # surely_outside descends every pair tree blindly instead of stopping where the upload step found nothing left to test
                       ("blind_prune_descent", dict(ACN_LANES="1", ACN_NO_PRUNE_LEVELS="1")),
                       ("all_generations", dict(ACN_LANES="1", ACN_WALK_PASSES="26", ACN_PRIVATE_LIMIT="0")),
                       ("three_passes", dict(ACN_LANES="1", ACN_WALK_PASSES="3", ACN_PRIVATE_LIMIT="1000", ACN_FETCH_WALK="512")),
                       ("all", dict(ACN_LANES="3", ACN_GRID="96", ACN_TEST_STACK_USE="3", ACN_PRIVATE_LIMIT="100000")),
                       # shading tasks above 32 samples on 64 lanes instead of 16 (k_shade<64>): another summation tree
                       # for a task's samples, so equal to rounding, not to the bit
                       ("wide_tasks", dict(ACN_LANES="1", ACN_CLASS0_MIN="32"))):
        for k in ("ACN_LANES", "ACN_GRID", "ACN_SHADE_GRID", "ACN_TEST_STACK_USE", "ACN_WALK_PASSES", "ACN_PRIVATE_LIMIT", "ACN_FETCH_WALK", "ACN_CLASS0_MIN", "ACN_NO_PRUNE_LEVELS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)                 # read by acn_scene_upload
        h = A.Handle(flat)
        frames[label] = (h.render_positions(pos, linear=True), h.last_stages())
        h.close()
    assert frames["all_private"][1]["private_rays"] == frames["plain"][1]["walk_rays"]
    assert frames["all_generations"][1]["private_rays"] < frames["plain"][1]["private_rays"] < frames["plain"][1]["walk_rays"]
    assert np.abs(frames["wide_tasks"][0] - frames["plain"][0]).max() <= 1e-11
    assert frames["wide_tasks"][1]["walk_rays"] == frames["plain"][1]["walk_rays"]
    for label in ("lanes", "small_grid", "stack_overflow", "all_private", "blind_prune_descent", "all_generations", "three_passes", "all"):
        assert np.array_equal(frames[label][0], frames["plain"][0]), label
        assert frames[label][1]["walk_rays"] == frames["plain"][1]["walk_rays"], label
        assert frames[label][1]["hard_rays"] == frames["plain"][1]["hard_rays"], label
    # one host synchronisation per chunk and lane, however many specular generations there are
    assert frames["plain"][1]["host_syncs"] == frames["plain"][1]["chunks"]
    sample = np.arange(0, len(pos), 37)
    cpu = oracle.render_positions(flat, pos[sample], linear=True)
    assert np.abs(frames["all"][0][sample] - cpu).max() <= TOL
