#!/usr/bin/env python3
"""bench.py -- Msamples/s (pixels x path_samples) of the trace/radiance path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one main pass (gradient_cycles = 0: one centre sample per pixel, /root/reference src/scene.c:1110-1119)
over the whole frame.  N > 1, two splits of the sample space (both behind the C ABI, include/actinon_hip.h):
  --split tiles    (default) every rank renders linear radiance for its interleaved tiles of 256 pixels
                   (acn_render_main_pass_shard_dev); ONE RCCL all_gather collects the ranks' parts -- every value is
                   copied, none added, so the frame is bit-identical to one GPU -- and rank 0 re-interleaves them;
  --split samples  every rank renders every pixel but only its share of the outermost direct-light and path sample loops
                   (ACN_SHARD_SAMPLES, src/scene.c:556,596); ONE RCCL all_reduce( sum, f64 ) of the per-pixel linear
                   radiance completes the frame: the split for a frame that is one expensive region.
Rank 0 then resolves (gamma + clamp + 8-bit pack, src/vectors.h:372-384, src/scene.c:76-82; always AFTER the exchange)
and copies the 8-bit image to the host.  The scene (flattened, device layout) is resident in HBM before the timed region;
pixel positions are generated on the device.

Workload (config.workload): wine_glass.acn at 1920x1080, path_samples 64, direct_samples 200 -- the scene and
sampling of BASELINE.json configs[1] at the resolution its metric is quoted on ("Msamples/s at 1920x1080").
`--workload c2` runs configs[1] verbatim (1280x720).

Rank 0 prints ONE JSON line.  `roofline` prices the main pass against HBM as BASELINE.json asks: algorithmic bytes =
24 B of radiance per pixel + 3 B of 8-bit image + one read of the flattened scene (SURVEY.md 8(d)'s floor; the path is
fp64-ALU bound, so frac is tiny by nature -- see DESIGN.md), with the per-kernel-family launch counts and average launch
durations of a one-lane pass beside it; `cpu_baseline` is the CPU oracle (a port of the reference's algorithm; the
reference itself needs the absent library beth) timed on the host cores on a strided pixel subset of the same frame, in
two builds: the deterministic one the parity tests use, and one compiled on this host with the reference's own flags
(-O2 -march=native, glibc libm, default contraction; makefile:10) -- `value` quotes the faster.
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch initialises HIP: the library's concurrent lanes need distinct hardware queues
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene builder, overrides)
    "wine_glass_1080p": ("wine_glass", dict(image_width=1920, image_height=1080, path_samples=64, direct_samples=200)),
    "c2": ("wine_glass", dict(image_width=1280, image_height=720, path_samples=64, direct_samples=200)),
    "c1": ("primitives", dict(image_width=400, image_height=300, path_samples=0, direct_samples=10)),
    "c4": ("diamond", dict(image_width=1920, image_height=1080, path_samples=512, direct_samples=50)),
    "c3": ("many_spheres:5:0", dict(image_width=1920, image_height=1080, path_samples=256, direct_samples=20)),
    "smoke": ("wine_glass", dict(image_width=160, image_height=90, path_samples=16, direct_samples=50)),
    # BASELINE.json configs[4]: the script's own settings (600x800, path 30 / direct 30); the scene comes from the
    # flattened fixture our interpreter produced from hanging_lamp.acn (tests/golden/make_scene_fixtures.py)
    "c5": ("fixture:hanging_lamp", dict()),
    # BASELINE.json configs[4] at its stated size and sampling.  One frame is ~3e12 rays (the two nested path levels make
    # the work quadratic in path_samples): a single GPU renders a strided pixel subset of the raster (--pixel-stride)
    "c5full": ("fixture:hanging_lamp", dict(image_width=3840, image_height=2160, path_samples=1024, direct_samples=30)),
    "paraffin_lamp": ("fixture:paraffin_lamp", dict()),
}

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6  # fp64 vector (non-MFMA) peak = half the guide's 157.3 TFLOP/s fp32 vector peak; counts an FMA as 2


def kernel_source_hash():
    """sha256 over the device sources the traffic profile belongs to (csrc/*.h, *.hip)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "actinon_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload, world):
    """HBM bytes per main pass from the rocprofv3 PMC passes committed under profiles/ (scripts/pmc_traffic.sh: FETCH_SIZE x2
    per the gfx950 correction + WRITE_SIZE, KB -> bytes, separate passes) for the same workload on one GPU -- but only if
    that profile was taken on THESE kernels (it is stamped with kernel_source_hash()); else None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        t = json.load(open(path)).get(workload, {})
        if t.get("kernel_source_hash") != kernel_source_hash():
            return None
        return t.get("hbm_bytes_per_step")
    except (OSError, ValueError):
        return None


def host_core_share():
    """Host cores this job may use: the cgroup CPU quota if one is set (a 1-GPU box gets a share of the host),
    else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("ACN_BENCH_CPU_THREADS")
    return int(env) if env else n


def native_oracle_build():
    """The oracle compiled on THIS host with the reference's own flags (/root/reference makefile:10: -O2 -march=native; glibc
    libm, the compiler's default floating-point contraction).  Returns the path of the shared object, or None when the host
    has no compiler.  Test infrastructure timed as the CPU baseline, never part of the product path."""
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        return None
    out = os.path.join(tempfile.mkdtemp(prefix="acn_oracle_native_"), "libacn_oracle_refflags.so")
    cmd = [cc, "-O2", "-march=native", "-fPIC", "-std=gnu11", "-DACN_ORACLE_LIBM", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oracle"), "-shared", "-o", out, os.path.join(ROOT, "oracle", "acn_oracle.c"), "-lm", "-lpthread"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    except (subprocess.SubprocessError, OSError):
        return None
    return out


def cpu_baseline(flat, width, height, path_samples, target_seconds=10.0, window=None):
    """CPU oracle on a strided pixel subset of the same frame (same scene, same sampling), all host cores, two builds.
    window = (x0, y0, w, h): time a centred sub-window instead (the heavy configs) and scale by pixel count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_binding import Oracle
    import actinon_amd as A
    cores = host_core_share()
    pos = A.main_pass_positions(width, height)
    if window:
        x0, y0, ww, wh = window
        pos = pos.reshape(height, width, 2)[y0:y0 + wh, x0:x0 + ww].reshape(-1, 2)
    n = pos.shape[0]
    unit_scale = max(path_samples, 1)
    builds = [("deterministic: oracle/libacn_oracle.so (software transcendentals of acn_detmath.h, -ffp-contract=off, "
               "-march=x86-64-v3, prebuilt) -- the build the parity tests compare with", Oracle())]
    native = native_oracle_build()
    if native:
        builds.append(("reference flags: -O2 -march=native, glibc libm, default contraction (makefile:10), compiled on this host",
                       Oracle(path=native)))
    results = []
    cnt = None
    for label, o in builds:
        # calibrate on a small strided probe (the oracle hands out 16 positions per lock, so no fewer than 64 per thread),
        # then size the sample for ~target_seconds
        probe_n = 4096 if path_samples < 256 else max(1024, 64 * cores)
        probe = pos[:: max(1, n // probe_n)]
        t0 = time.perf_counter()
        o.render_positions(flat, probe, linear=True, threads=cores)
        dt = max(time.perf_counter() - t0, 1e-3)
        rate = probe.shape[0] / dt
        want = int(min(n, max(probe.shape[0], rate * target_seconds)))
        stride = max(1, n // want)
        sample = pos[::stride]
        t0 = time.perf_counter()
        o.render_positions(flat, sample, linear=True, threads=cores)
        dt = time.perf_counter() - t0
        results.append({"build": label, "value": sample.shape[0] * unit_scale / dt / 1e6, "pixels": int(sample.shape[0]),
                        "stride": stride, "seconds": dt})
        if cnt is None:
            # the reference algorithm's fp64 work per pixel (cost table of SURVEY.md App. B), counted on the probe
            _, cnt = o.render_positions(flat, probe, linear=True, threads=cores, counters=True)
            probe_pixels = probe.shape[0]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        model = "unknown"
    best = max(results, key=lambda r: r["value"])
    where = (f"the {window[2]}x{window[3]} window at ({window[0]}, {window[1]}) of the {width}x{height} frame, scaled by pixel count "
             f"(BASELINE.md 2: extrapolated)" if window else f"the {width}x{height} frame")
    return {
        "value": best["value"],
        "unit": "Msamples/s" if path_samples else "Mpixels/s",
        "cores": cores,
        "kind": "port",
        "cpu_model": model,
        "quoted_build": best["build"],
        "builds": results,
        "flop_per_pixel": cnt["flop"] / probe_pixels,
        "transcendentals_per_pixel": cnt["transc"] / probe_pixels,
        "sample": f"every {best['stride']}th pixel of {where} ({best['pixels']} pixels x {path_samples} path samples, "
                  f"{best['seconds']:.1f} s, {cores} threads; every build timed on its own ~{target_seconds:.0f} s sample)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wine_glass_1080p", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true",
                    help="heavy workloads: no extra passes for per-stage events and work counters after the timed steps")
    ap.add_argument("--pixel-stride", type=int, default=1,
                    help="render every K-th pixel of the raster only (heavy configs on one GPU); value counts the pixels rendered")
    ap.add_argument("--rows", default=None,
                    help="render only the raster rows Y0:Y1 (a frame too long for one call is rendered in bands; value counts the pixels rendered)")
    ap.add_argument("--save-image", default=None, help="write the last frame as PNM (rank 0)")
    ap.add_argument("--split", default="tiles", choices=["tiles", "samples"],
                    help="N > 1: pixel tiles + all_gather (bit-identical to one GPU), or sample sub-ranges of every pixel + all_reduce( sum )")
    ap.add_argument("--cpu-window", default=None,
                    help="CPU baseline on a centred WxH sub-window of the frame, scaled by pixel count (BASELINE.md 2; heavy configs), e.g. 240x135")
    ap.add_argument("--checksum", default=None,
                    help="write per-tile fixed-point sums of the last linear frame to this JSON file and compare with tests/golden/frame_checksums.json")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import actinon_amd as A
    from actinon_amd import dist as adist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # ACN_BENCH_SINGLE_DEVICE=1: rehearsal of the N > 1 code path on a 1-GPU box (every rank on cuda:0, gloo
    # all-reduce through host memory). Never used for reported numbers.
    rehearsal = os.environ.get("ACN_BENCH_SINGLE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    builder, ov = WORKLOADS[args.workload]
    if builder.startswith("fixture:"):
        flat = A.Flat.load(os.path.join(ROOT, "tests", "golden", "scenes", builder.split(":")[1] + ".npz"), **ov)
    else:
        scene = A.Scene.build(builder, **ov)
        flat = scene.flatten()
    W, H, S = int(flat.params.image_width), int(flat.params.image_height), int(flat.params.path_samples)
    n_pix = W * H
    first_pix = 0
    if args.rows:
        if world != 1 or args.pixel_stride > 1:
            raise SystemExit("--rows is a single-GPU option without --pixel-stride")
        y0, y1 = (int(v) for v in args.rows.split(":"))
        if not 0 <= y0 < y1 <= H:
            raise SystemExit("--rows outside the raster")
        first_pix, n_pix = y0 * W, (y1 - y0) * W
    handle = A.Handle(flat, device=local_rank)

    # resident buffers.  N > 1: the frame is cut into tiles of 256 pixels dealt round-robin to the ranks (acn_shard_tile_*);
    # a rank renders its tiles into a compact part, the parts are all-gathered over RCCL (every value is copied, none
    # added: bit-identical to one GPU, 1 / N of the frame sent per rank) and rank 0 interleaves, resolves and copies out
    if args.pixel_stride > 1:
        if world != 1:
            raise SystemExit("--pixel-stride is a single-GPU option")
        idx_np = np.arange(0, n_pix, args.pixel_stride, dtype=np.int64)
        n_pix = idx_np.shape[0]
        pos = torch.from_numpy(adist.pixel_positions(idx_np, W)).to(dev)
    else:
        pos = None
    by_samples = world > 1 and args.split == "samples"
    n_rank = n_pix if by_samples else adist.rank_count(n_pix, rank, world)
    padded = adist.padded(n_pix, world)
    part = torch.empty((padded, 3), dtype=torch.float64, device=dev) if not by_samples else None
    gathered = torch.empty((world * padded, 3), dtype=torch.float64, device=dev) if world > 1 and not by_samples else None
    if by_samples:
        handle.sample_shard = (rank, world)
    frame = torch.zeros((n_pix, 3), dtype=torch.float64, device=dev)
    rgb8 = torch.empty((n_pix, 3), dtype=torch.uint8, device=dev)
    host_img = torch.empty((n_pix, 3), dtype=torch.uint8).pin_memory()
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms = []

    def step(record):
        if by_samples:
            # every rank: all pixels, its share of the outermost sample loops; linear partial radiance, summed over the ranks
            handle.render_main_pass_dev(0, n_pix, frame.data_ptr(), linear=True, stream=stream)
            if rehearsal:
                host = frame.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                frame.copy_(host)
            else:
                dist.all_reduce(frame, op=dist.ReduceOp.SUM)
            if rank == 0:
                handle.resolve_dev(frame.data_ptr(), n_pix, None, rgb8.data_ptr(), stream=stream)
                host_img.copy_(rgb8, non_blocking=True)
        elif world > 1:
            handle.render_main_pass_shard_dev(0, n_pix, rank, world, part.data_ptr(), linear=True, stream=stream)
            if rehearsal:
                host = gathered.cpu()
                dist.all_gather_into_tensor(host, part.cpu())
                gathered.copy_(host)
            else:
                dist.all_gather_into_tensor(gathered, part)
            if rank == 0:
                handle.shard_unpack_dev(gathered.data_ptr(), n_pix, world, frame.data_ptr(), stream=stream)
                handle.resolve_dev(frame.data_ptr(), n_pix, None, rgb8.data_ptr(), stream=stream)
                host_img.copy_(rgb8, non_blocking=True)
        else:
            if pos is None:
                handle.render_main_pass_dev(first_pix, n_pix, frame.data_ptr(), linear=True, stream=stream)
            else:
                handle.render_positions_dev(pos.data_ptr(), pos.shape[0], frame.data_ptr(), linear=True, stream=stream)
            handle.resolve_dev(frame.data_ptr(), n_pix, None, rgb8.data_ptr(), stream=stream)
            host_img.copy_(rgb8, non_blocking=True)
        if record:
            torch.cuda.synchronize()
            kernel_ms.append(handle.last_kernel_ms())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # device time of the main pass: HIP events recorded by the library on its launch streams, of the LAST TIMED step (the
    # passes below carry per-launch events or work counters and are not what `value` timed)
    families = None
    k_timed = handle.last_kernel_ms()
    if args.quick:
        stages = handle.last_stages()
        counters = None
    else:
        handle.stage_timing = True          # per-launch events only here: they cost ~0.7 % of a frame
        for _ in range(min(2, max(1, args.steps))):
            step(True)
        stages = handle.last_stages()
        handle.stage_timing = False
        # one extra, untimed pass through the instrumented kernels for the work counters (same traversal as the timed
        # kernels: the instrumented variants carry the prune programs too)
        handle.count_work = True
        step(False)
        torch.cuda.synchronize()
        counters = handle.last_counters()
        handle.count_work = False
        if world == 1:
            # per kernel family, cleanly: the same pass on ONE lane (no overlapping lanes), per-launch HIP events
            lanes_before = os.environ.get("ACN_LANES")
            os.environ["ACN_LANES"] = "1"
            h1 = A.Handle(flat, device=local_rank)
            if lanes_before is None:
                del os.environ["ACN_LANES"]
            else:
                os.environ["ACN_LANES"] = lanes_before
            for timing in (False, True):
                h1.stage_timing = timing
                if pos is None:
                    h1.render_main_pass_dev(first_pix, n_pix, frame.data_ptr(), linear=True, stream=stream)
                else:
                    h1.render_positions_dev(pos.data_ptr(), pos.shape[0], frame.data_ptr(), linear=True, stream=stream)
                torch.cuda.synchronize()
            s1 = h1.last_stages()
            h1.close()
            families = {}
            for fam, ms_key, n_key in (("k_walk (+ k_shade_hits)", "walk_ms", "walk_launches"), ("k_shade<64|16|4|1>", "shade_ms", "shade_launches"),
                                       ("k_hard_shadow + k_hard_path", "hard_ms", "hard_launches"), ("k_finalize", "finalize_ms", "finalize_launches")):
                families[fam] = {"ms_per_pass": s1[ms_key], "launches": int(s1[n_key]),
                                 "avg_launch_ms": s1[ms_key] / max(1.0, s1[n_key])}
            families["pass_total_ms_one_lane"] = s1["total_ms"]
    k_ms = float(k_timed)

    check = None
    if rank == 0 and args.checksum:
        # per-tile fixed-point sums of the linear frame of the last step (pixel sums are 2^-40 fixed point and order
        # independent: a frame is reproducible bit for bit, so the digest pins the WHOLE frame, not a pixel subset)
        import hashlib
        lin = frame.cpu().numpy()
        fx = np.rint(lin * 1099511627776.0).astype(np.int64)
        tile = 65536
        sums = [[int(v) for v in fx[i:i + tile].sum(axis=0)] for i in range(0, fx.shape[0], tile)]
        digest = hashlib.sha256(lin.tobytes()).hexdigest()
        key = f"{args.workload}/stride{args.pixel_stride}" + (f"/rows{args.rows}" if args.rows else "")
        check = {"key": key, "sha256": digest, "tile_pixels": tile, "tile_sums_fixed_2^-40": sums}
        gpath = os.path.join(ROOT, "tests", "golden", "frame_checksums.json")
        golden = json.load(open(gpath)) if os.path.exists(gpath) else {}
        if key in golden:
            check["golden"] = "match" if golden[key]["sha256"] == digest else "MISMATCH"
            if check["golden"] != "match":
                bad = [i for i, (a, b) in enumerate(zip(sums, golden[key]["tile_sums_fixed_2^-40"])) if a != b]
                check["tiles_that_differ"] = bad[:32]
        else:
            check["golden"] = "no entry"
        with open(args.checksum, "w") as f:
            json.dump({key: {k: check[k] for k in ("sha256", "tile_pixels", "tile_sums_fixed_2^-40")}}, f)

    if rank == 0:
        unit = max(S, 1)
        value = n_pix * unit * args.steps / elapsed / 1e6
        # Algorithmic HBM bytes of one main pass on this rank: the radiance written once (24 B per position;
        # positions are generated on the device), the 8-bit image, one read of the flattened scene.  (Rounds 1-2 also
        # charged a scene read per workgroup and LAUNCH, which moved with the launch count; the scene is 9 KB .. 7 MB and
        # lives in the scalar cache / LDS / L2.)
        scene_bytes = flat.n_nodes * 288 + flat.c.n_elems * 8      # GNode 192 B + GMat 96 B per node, elems twice
        alg_bytes = n_rank * 24 + (n_pix * 3 if rank == 0 else 0) + scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        dominant = None
        if families:
            fam = max((k for k in families if isinstance(families[k], dict)), key=lambda k: families[k]["ms_per_pass"])
            dominant = dict(families[fam], family=fam, share_of_one_lane_pass=families[fam]["ms_per_pass"] / max(families["pass_total_ms_one_lane"], 1e-9))
        partition = (f"sample sub-ranges of every pixel over {world} rank(s) (ACN_SHARD_SAMPLES)" if by_samples
                     else f"pixel tiles of {adist.TILE}, round-robin over {world} rank(s) (acn_shard_tile_*)")
        exchange = "none" if world == 1 else (f"RCCL all_reduce( sum, f64 ) of the linear frame ({n_pix * 24} B)" if by_samples
                                              else f"RCCL all_gather of the ranks' parts ({padded * 24} B each), no reduction")
        out = {
            "metric": "Msamples/s (pixels x path_samples), one main pass" if S else "Mpixels/s (path_samples = 0), one main pass",
            "value": value,
            "unit": "Msamples/s" if S else "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (single-device rehearsal of the multi-rank path, not a measurement)" if rehearsal else ""),
            "config": {"workload": f"{builder} {W}x{H} path_samples={S} direct_samples={int(flat.params.direct_samples)} "
                                   f"trace_depth={int(flat.params.trace_depth)} (BASELINE.json configs[1] scene+sampling"
                                   f"{' at the metric resolution 1920x1080' if args.workload == 'wine_glass_1080p' else ''})"
                       if builder == "wine_glass" else f"{builder} {W}x{H} path_samples={S} direct_samples={int(flat.params.direct_samples)}",
                       "pixels": n_pix, "path_samples": S,
                       "pixel_subset": (f"every {args.pixel_stride}th pixel of the {W}x{H} raster" if args.pixel_stride > 1 else
                                        f"rows {args.rows} of the {W}x{H} raster" if args.rows else "all"),
                       "split": args.split if world > 1 else "none", "partition": partition, "exchange": exchange,
                       # what torch.distributed actually ran on (N > 1): "nccl" is RCCL on ROCm; world size as the process group saw it
                       "backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1),
                       "parallelism": (f"{args.split}{world}" if world > 1 else "none")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, world),
                         "kernel": "one main pass = the launch chain k_walk passes / k_shade x 4 size classes / k_hard_shadow / k_hard_path / "
                                   "k_shade_hits per path level + k_finalize; HIP events on the launch streams around the pass",
                         "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes,
                         "algorithmic_bytes_are": "24 B radiance per position + 3 B 8-bit pixel + one read of the flattened scene (SURVEY.md 8(d) floor)",
                         "dominant_kernel": dominant, "kernel_families_one_lane": families,
                         "family_ms_summed_over_lanes": {"k_walk+k_shade_hits": stages["walk_ms"], "k_shade": stages["shade_ms"],
                                                         "k_hard_*": stages["hard_ms"], "k_finalize": stages["finalize_ms"]},
                         "workspace_bytes": stages.get("workspace_bytes"),
                         "traffic_profile": "profiles/traffic.json (stamped with the kernel source hash; null when stale)",
                         "note": "the path is fp64 arithmetic with divergent CSG traversal, bound by latency (DESIGN.md 4d); the HBM roofline is "
                                 "reported because BASELINE.json asks for it (DESIGN.md 7)"},
            "roofline_fp64": None if counters is None else {
                "bound": "fp64 vector ALU (no MFMA: the path has no dense contraction)",
                "flop": counters["flop"], "transcendentals": counters["transcendentals"],
                "achieved": counters["flop"] / (k_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": counters["flop"] / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "note": "flop = events the instrumented kernels executed on rank 0 x the unit costs of SURVEY.md App. B "
                        "(actinon_amd/csrc/acn_costs.h; a transcendental call is tallied separately, not as flops); "
                        "cpu_baseline.flop_per_pixel is the same tally of the reference's algorithm by the oracle"},
            "stages": stages,
            "work": None if counters is None else {
                "rays_per_step_rank0": counters["trans_rays"] + counters["shadow_rays"],
                "obj_hit_tests_rank0": counters["obj_hits"],
                "grays_per_s_rank0": (counters["trans_rays"] + counters["shadow_rays"]) / (k_ms * 1e-3) / 1e9},
            "frame_check": check if check is None else {k: check[k] for k in check if k != "tile_sums_fixed_2^-40"},
        }
        if world == 1 and not args.no_cpu_baseline:
            window = None
            if args.cpu_window:
                ww, wh = (int(v) for v in args.cpu_window.lower().split("x"))
                window = ((W - ww) // 2, (H - wh) // 2, ww, wh)
            out["cpu_baseline"] = cpu_baseline(flat, W, H, S, window=window)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
        if args.save_image:
            from actinon_amd._lib import host
            img = host_img.numpy().astype(np.float64) / 256.0 + 0.5 / 256.0
            if args.pixel_stride == 1:
                host.acn_write_pnm(args.save_image.encode(), img.ctypes.data, W, n_pix // W)
            else:   # every k-th pixel of the raster: not an image any more, one row of the pixels that were rendered
                host.acn_write_pnm(args.save_image.encode(), img.ctypes.data, img.shape[0], 1)

    handle.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
