"""BASELINE.json configs[2..4] at their STATED size and sampling on the GPU, against the CPU oracle.

The oracle needs hours for a whole frame at these settings (the reference's radiance recursion nests two path levels,
src/scene.c:584-621: every first-level path sample whose hit is diffuse starts another path_samples * intensity
loop), so parity is checked on a strided subset of the full raster: the camera, the scene, path_samples, direct_samples,
trace_depth and the pixel -> ray mapping are the config's own, only the number of pixels looked at is reduced.  That puts
the code paths only these configs reach under the oracle: the 5-level / 37 449-node many_spheres scene that does not fit
LDS, 8- and 16-iteration sample loops of the 64-lane shading kernel, thousands of path-sample hits per pixel, and the
chunking of a call.

Also here: the queue-overflow retry path (a chunk whose records do not fit is halved and redone) and the cancel flag of
acn_render_opts inside the library.
"""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

import actinon_amd as A
import scenes_util as S

pytestmark = pytest.mark.gpu
TOL = 1e-9
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (builder, overrides, pixels looked at)
CONFIGS = {
    # many_spheres.acn:23-43,62-98: 8^5 spheres under five levels of enveloped compounds, auto envelopes from the
    # Monte-Carlo estimator (run on the GPU, as `many_spheres:5:0` asks)
    "c3_many_spheres": ("many_spheres:5:0", dict(image_width=1920, image_height=1080, path_samples=256, direct_samples=20), 2048),
    # diamond.acn:23-43,194-212
    "c4_diamond": ("diamond", dict(image_width=1920, image_height=1080, path_samples=512, direct_samples=50), 2048),
    # hanging_lamp/hanging_lamp.acn:26-47 through the flattened fixture (2 847 nodes)
    "c5_hanging_lamp": ("fixture:hanging_lamp", dict(image_width=3840, image_height=2160, path_samples=1024, direct_samples=30), 1024),
}


def load(builder, ov):
    if builder.startswith("fixture:"):
        return A.Flat.load(os.path.join(HERE, "golden", "scenes", builder.split(":")[1] + ".npz"), **ov)
    sc = A.Scene.build(builder, **ov)
    return sc.flatten()


def strided_positions(flat, count):
    """`count` pixel centres spread over the whole raster: every (W*H // count)-th pixel in raster order, the stride
    made odd so that the columns drift from row to row."""
    w, h = int(flat.params.image_width), int(flat.params.image_height)
    stride = w * h // count
    stride -= 1 - stride % 2
    idx = np.arange(0, w * h, stride)[:count]
    return A.main_pass_positions(w, h)[idx]


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert A.device_count() >= 1, "no HIP device: the gpu tests must run on the GPU box"


@pytest.mark.parametrize("wide", [False, True], ids=["tasks_on_16_lanes", "tasks_on_64_lanes"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_at_stated_size_matches_oracle_on_strided_pixels(oracle, name, wide, monkeypatch):
    """Both widths of a shading task (size_class in acn_pipeline.h): a whole wavefront per point above 32 samples -- 8- and
    16-round sample loops of k_shade<64> at these configs' 256 - 1024 path samples, the library's choice for these scenes --
    and 16 lanes per point whatever the count."""
    monkeypatch.setenv("ACN_CLASS0_MIN", "32" if wide else "1000000")
    builder, ov, count = CONFIGS[name]
    flat = load(builder, ov)
    assert int(flat.params.path_samples) == ov["path_samples"] and int(flat.params.image_width) == ov["image_width"]
    pos = strided_positions(flat, count)
    assert len(pos) == count
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    st = h.last_stages()
    h.close()
    t0 = time.time()
    cpu = oracle.render_positions(flat, pos, linear=True)
    print(f"{name}: {count} pixels, gpu {st['total_ms']:.0f} ms ({st['chunks']:.0f} chunks, {st['levels']:.0f} levels, "
          f"peak children {st['peak_children']:.0f}), oracle {time.time() - t0:.1f} s")
    err = np.abs(gpu - cpu)
    assert err.max() <= TOL, f"{name}: {(err > TOL).any(axis=1).sum()} of {count} pixels differ, max {err.max():.3e}"
    assert np.array_equal(A.cps_from_cl(np.clip(gpu, 0, None) ** flat.params.gamma), A.cps_from_cl(np.clip(cpu, 0, None) ** flat.params.gamma))
    # the subset is not degenerate: it sees objects, not only background
    assert gpu.std() > 1e-3


def test_queue_overflow_retry_is_bit_identical(oracle, monkeypatch):
    """A chunk whose path-sample hits do not fit the queues is halved and redone (launch_render in actinon_hip.hip).
    Forced here with a small workspace and a chunk that is too large for it; the image must not change by a bit."""
    sc = A.Scene.build("wine_glass", image_width=320, image_height=180, path_samples=64, direct_samples=50)
    flat = sc.flatten()
    pos = S.positions(flat)
    monkeypatch.setenv("ACN_LANES", "1")
    h = A.Handle(flat)
    ref = h.render_positions(pos, linear=True)
    st_ref = h.last_stages()
    h.close()
    assert st_ref["retries"] == 0
    monkeypatch.setenv("ACN_WORKSPACE_MB", "48")          # 65 536 records per queue (the floor)
    monkeypatch.setenv("ACN_CHUNK", str(len(pos)))        # every position in one chunk: ~3 hits per pixel do not fit
    h = A.Handle(flat)
    forced = h.render_positions(pos, linear=True)
    st = h.last_stages()
    h.close()
    assert st["retries"] > 0 and st["chunks"] > 1, st
    assert np.array_equal(forced, ref)
    # ... and through the concurrent lanes, each with its share of the small workspace
    monkeypatch.setenv("ACN_LANES", "2")
    monkeypatch.setenv("ACN_WORKSPACE_MB", "96")
    h = A.Handle(flat)
    forced4 = h.render_positions(pos, linear=True)
    st4 = h.last_stages()
    h.close()
    assert st4["retries"] > 0, st4
    assert np.array_equal(forced4, ref)
    sample = np.arange(0, len(pos), 53)
    cpu = oracle.render_positions(flat, pos[sample], linear=True)
    assert np.abs(forced[sample] - cpu).max() <= TOL


def test_cancel_flag_inside_the_library(monkeypatch):
    """acn_render_opts.cancel is the SIGINT flag of src/scene.c:893,978: set before the call nothing is rendered; set
    while the call is running the library stops between two chunks and returns ACN_ERR_CANCELLED."""
    sc = A.Scene.build("diamond", image_width=256, image_height=256, path_samples=256, direct_samples=50)
    flat = sc.flatten()
    pos = S.positions(flat)
    monkeypatch.setenv("ACN_CHUNK", "512")                # many chunks: many polls
    for lanes in ("1", "4"):
        monkeypatch.setenv("ACN_LANES", lanes)
        h = A.Handle(flat)
        h.cancel = C.c_int(1)
        with pytest.raises(A.AcnError) as e:
            h.render_positions(pos)
        assert e.value.status == A.abi.ACN_ERR_CANCELLED
        h.cancel = C.c_int(0)
        flag = h.cancel
        timer = threading.Timer(0.05, lambda: setattr(flag, "value", 1))
        t0 = time.time()
        timer.start()
        with pytest.raises(A.AcnError) as e:
            h.render_positions(pos)
        dt = time.time() - t0
        timer.cancel()
        assert e.value.status == A.abi.ACN_ERR_CANCELLED
        assert dt < 30, dt
        # the handle stays usable
        h.cancel = None
        small = h.render_positions(pos[:512], linear=True)
        assert np.isfinite(small).all()
        h.close()


def test_sample_shards_sum_to_the_unsharded_render(oracle):
    """ACN_SHARD_SAMPLES on one GPU: the shares of ranks 0 .. world-1 rendered one after the other add up to the unsharded
    linear radiance (fixed-point pixel sums: the only difference is where partial sums were rounded to 2^-40), every share
    equals the oracle's share, and only rank 0 carries the terms under no sample loop."""
    sc = A.Scene.build("wine_glass", image_width=96, image_height=54, path_samples=64, direct_samples=200)
    flat = sc.flatten()
    pos = S.positions(flat)
    h = A.Handle(flat)
    full = h.render_positions(pos, linear=True)
    for world in (2, 8):
        total = np.zeros_like(full)
        for rank in range(world):
            h.sample_shard = (rank, world)
            part = h.render_positions(pos, linear=True)
            if world == 2:
                cpu = oracle.render_positions(flat, pos, linear=True, shard=(rank, world))
                assert np.abs(part - cpu).max() <= TOL, (rank, np.abs(part - cpu).max())
            total += part
        h.sample_shard = None
        assert np.abs(total - full).max() <= 1e-10, (world, np.abs(total - full).max())
    # a camera ray that hits nothing is background on rank 0 and nothing on the others
    sky = np.array([[48.5, 0.5]])
    h.sample_shard = (0, 2)
    a = h.render_positions(sky, linear=True)
    h.sample_shard = (1, 2)
    b = h.render_positions(sky, linear=True)
    h.sample_shard = None
    assert np.allclose(a + b, h.render_positions(sky, linear=True), atol=1e-11)
    h.close()


def test_tile_shards_reassemble_bit_for_bit():
    """acn_render_main_pass_shard_dev + acn_shard_unpack_dev: the parts of the ranks, gathered rank-major, give the frame of
    acn_render_main_pass_dev bit for bit (ragged frame: the last tile is short, one rank has a tile less)."""
    import torch
    sc = A.Scene.build("wine_glass", image_width=100, image_height=77, path_samples=16, direct_samples=50)
    flat = sc.flatten()
    n = 100 * 77
    h = A.Handle(flat)
    ref = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
    h.render_main_pass_dev(0, n, ref.data_ptr(), linear=True)
    for world in (3, 8):
        padded = A.hip.acn_shard_tile_padded(n, world)
        gathered = torch.full((world, padded, 3), -1.0, dtype=torch.float64, device="cuda:0")
        for rank in range(world):
            h.render_main_pass_shard_dev(0, n, rank, world, gathered[rank].data_ptr(), linear=True)
        frame = torch.empty((n, 3), dtype=torch.float64, device="cuda:0")
        h.shard_unpack_dev(gathered.data_ptr(), n, world, frame.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(frame, ref), world
        # padding rows are zero, never garbage
        for rank in range(world):
            cnt = A.hip.acn_shard_tile_count(n, rank, world)
            assert (gathered[rank, cnt:] == 0).all()
    h.close()


def test_bench_two_ranks_on_one_gpu_match_one_rank(tmp_path):
    """bench.py's N > 1 path end to end on real hardware: two processes (ranks) share the one GPU of the box
    (ACN_BENCH_SINGLE_DEVICE=1: gloo instead of RCCL between them), each renders its tiles through
    acn_render_main_pass_shard_dev, the parts are all-gathered, rank 0 unpacks, resolves and writes the image -- which must
    be the image of a one-rank run, byte for byte."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ACN_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = tmp_path / "two.pnm"
    one = tmp_path / "one.pnm"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "smoke", "--no-cpu-baseline", "--save-image", str(two)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "all_gather" in json.dumps(line["config"])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "smoke",
                        "--no-cpu-baseline", "--save-image", str(one)], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert two.read_bytes() == one.read_bytes()


def test_bench_two_ranks_sample_split_on_one_gpu(tmp_path):
    """bench.py --split samples end to end: two ranks on the one GPU of the box (gloo between them), each renders EVERY pixel
    but only its half of the outermost sample loops (ACN_SHARD_SAMPLES), the linear frames are sum-reduced, rank 0 resolves
    after the reduce.  The partial sums are rounded to 2^-40 per rank instead of once, so the linear frame differs from a
    one-rank run by <= 1e-10 (test_sample_shards_sum_to_the_unsharded_render); in the 8-bit image that can move a value
    sitting on a quantisation boundary by one step, nothing more."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ACN_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = tmp_path / "two.pnm"
    one = tmp_path / "one.pnm"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "smoke", "--split", "samples", "--no-cpu-baseline", "--save-image", str(two)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["config"]["split"] == "samples" and "all_reduce" in line["config"]["exchange"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "smoke",
                        "--no-cpu-baseline", "--save-image", str(one)], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a = np.frombuffer(two.read_bytes(), dtype=np.uint8).astype(np.int16)
    b = np.frombuffer(one.read_bytes(), dtype=np.uint8).astype(np.int16)
    assert a.shape == b.shape
    d = np.abs(a - b)
    assert d.max() <= 1 and ( d != 0 ).mean() < 1e-3, ( d.max(), ( d != 0 ).mean() )


@pytest.mark.parametrize("args", [[], ["--workload", "c2"], ["--workload", "c1"], ["--workload", "c5"], ["--workload", "paraffin_lamp"],
                                  ["--workload", "c4", "--pixel-stride", "64"], ["--workload", "c3", "--pixel-stride", "64"]],
                         ids=["wine_glass_1080p", "c2", "c1", "hanging_lamp_600x800", "paraffin_lamp", "c4_every_64th", "c3_every_64th"])
def test_whole_frame_equals_the_committed_digest(tmp_path, args):
    """Whole frames rendered through bench.py and compared with tests/golden/frame_checksums.json: sha256 of the linear f64
    frame and per-tile fixed-point sums -- the headline frame (wine_glass 1920 x 1080, path 64 / direct 200: 2 073 600 pixels),
    BASELINE configs C1 / C2, the two lamp scenes at their script sizes and every 64th pixel of C3 / C4 at stated size.  Pixel
    sums are order-independent 2^-40 fixed point, so a frame is reproducible bit for bit across lanes, chunkings, boxes and
    kernel builds; the 1080p digest was taken in round 3 and did not change through that round's kernel changes (cone culling,
    merged acos ranges, reservation prefetch).  The oracle side of these frames: tests/test_gpu_parity.py and
    test_config_at_stated_size_matches_oracle_on_strided_pixels."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    out = tmp_path / "digest.json"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args, "--steps", "1", "--warmup", "0", "--quick", "--no-cpu-baseline",
                        "--checksum", str(out)], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["frame_check"].get("golden") == "match", line["frame_check"]


@pytest.mark.parametrize("name,ov", [
    ("wine_glass_1080p", dict(image_width=1920, image_height=1080, path_samples=64, direct_samples=200)),
    ("c2", dict(image_width=1280, image_height=720, path_samples=64, direct_samples=200)),
], ids=["wine_glass_1080p", "c2_1280x720"])
def test_whole_frame_matches_oracle_on_every_pixel(oracle, name, ov):
    """The headline frame of bench.py (wine_glass 1920 x 1080, path 64 / direct 200: 2 073 600 pixels) and BASELINE configs[1]
    (1280 x 720) rendered by the GPU AND by the CPU oracle, every pixel compared: max |delta| <= 1e-9 per channel on the linear
    radiance, and the same 8-bit image after cl_s_sat / cps_from_cl (src/vectors.h:372-384, src/scene.c:76-82).  The main-pass
    positions are those of src/scene.c:1110-1119 (pixel centres, raster order).  The oracle needs ~7 s / ~3 s on the 16
    threads of the GPU box.  (The committed digests guard against regressions between GPU builds; this is the parity check.)"""
    sc = A.Scene.build("wine_glass", **ov)
    flat = sc.flatten()
    w, hh = int(flat.params.image_width), int(flat.params.image_height)
    assert (w, hh) == (ov["image_width"], ov["image_height"])
    pos = A.main_pass_positions(w, hh)
    h = A.Handle(flat)
    gpu = h.render_positions(pos, linear=True)
    st = h.last_stages()
    h.close()
    t0 = time.time()
    cpu = oracle.render_positions(flat, pos, linear=True)
    t_cpu = time.time() - t0
    err = np.abs(gpu - cpu)
    worst = int(err.max(axis=1).argmax())
    print(f"{name}: {len(pos)} pixels, gpu {st['total_ms']:.1f} ms, oracle {t_cpu:.1f} s, max |gpu - oracle| {err.max():.3e} "
          f"at pixel {worst} ({worst % w}, {worst // w}), mean {err.mean():.3e}")
    assert err.max() <= TOL, f"{name}: {(err > TOL).any(axis=1).sum()} of {len(pos)} pixels differ, max {err.max():.3e}"
    g8 = A.cps_from_cl(np.clip(gpu, 0, None) ** flat.params.gamma)
    c8 = A.cps_from_cl(np.clip(cpu, 0, None) ** flat.params.gamma)
    # A pixel sum is a 2^-40 fixed-point number on the GPU and a chain of double additions in the oracle; where a channel sits
    # within 1e-12 of an 8-bit rounding boundary the two may fall on different sides of it: at most a handful of the 6.2 M values,
    # never by more than one step
    diff = np.abs(g8.astype(np.int16) - c8.astype(np.int16))
    assert diff.max() <= 1 and int((diff != 0).sum()) <= 8, (int(diff.max()), int((diff != 0).sum()))


def test_cold_handle_learns_from_a_sample_and_renders_the_same_bits(oracle, monkeypatch):
    """A cold handle renders a strided SAMPLE of the call once, throws it away and sizes its queues from the exact record counts
    (learn_rates in actinon_hip.hip): the first frame must come out bit for bit like a later one and like a frame rendered without
    the learning pass, no chunk may have to be redone, and the sampled pixels -- whose accumulators the learning pass touched and
    cleared -- must equal the oracle's."""
    sc = A.Scene.build("wine_glass", image_width=320, image_height=180, path_samples=64, direct_samples=50)
    flat = sc.flatten()
    pos = S.positions(flat)
    assert len(pos) >= 16384          # large enough for the learning pass
    h = A.Handle(flat)
    first = h.render_positions(pos, linear=True)
    st_first = h.last_stages()
    second = h.render_positions(pos, linear=True)
    st_second = h.last_stages()
    h.close()
    assert st_first["retries"] == 0 and st_second["retries"] == 0, (st_first, st_second)
    assert np.array_equal(first, second)
    monkeypatch.setenv("ACN_LEARN_SAMPLE", "0")
    h = A.Handle(flat)
    plain = h.render_positions(pos, linear=True)
    h.close()
    assert np.array_equal(first, plain)
    # the lanes of a cold handle are made on a helper thread beside the learning pass (render_lanes / lane_objects); made before
    # it (ACN_COLD_PIPELINE=0), or added by a later, larger call of a handle whose first call was small, they render the same bits
    monkeypatch.delenv("ACN_LEARN_SAMPLE")
    monkeypatch.setenv("ACN_COLD_PIPELINE", "0")
    h = A.Handle(flat)
    before = h.render_positions(pos, linear=True)
    h.close()
    assert np.array_equal(first, before)
    monkeypatch.delenv("ACN_COLD_PIPELINE")
    h = A.Handle(flat)
    small = h.render_positions(pos[:20000], linear=True)          # one lane: the handle itself learns the rates
    chunks_small = h.last_stages()["chunks"]
    grown = h.render_positions(pos, linear=True)                  # several lanes, made now, inherit them
    chunks_grown = h.last_stages()["chunks"]
    h.close()
    assert np.array_equal(first[:20000], small) and np.array_equal(first, grown)
    assert chunks_small == 1 and chunks_grown > 1, (chunks_small, chunks_grown)     # (every lane of a call runs at least one chunk)
    # lanes made during acn_scene_upload (ACN_EARLY_LANES=1, early_lanes_begin): adopted by the first call on lanes, unused by a
    # one-lane call, freed with a handle that never rendered
    monkeypatch.setenv("ACN_EARLY_LANES", "1")
    A.Handle(flat).close()
    h = A.Handle(flat)
    small_early = h.render_positions(pos[:20000], linear=True)
    early = h.render_positions(pos, linear=True)
    st_early = h.last_stages()
    h.close()
    monkeypatch.delenv("ACN_EARLY_LANES")
    assert np.array_equal(first[:20000], small_early) and np.array_equal(first, early)
    assert st_early["chunks"] > 1 and st_early["retries"] == 0, st_early
    stride =len(pos) // min(4096, len(pos) // 4)      # the positions the learning pass rendered and cleared
    sample = np.arange(0, len(pos), stride)[:512]
    cpu = oracle.render_positions(flat, pos[sample], linear=True)
    assert np.abs(first[sample] - cpu).max() <= TOL
