#!/usr/bin/env python3
"""bench.py -- Msamples/s (pixels x path_samples) of the trace/radiance path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one main pass (gradient_cycles = 0: one centre sample per pixel, /root/reference src/scene.c:1110-1119)
over the whole frame: every rank renders linear radiance for its interleaved pixel tiles (actinon_amd/dist.py),
the frame is completed by one RCCL sum all-reduce of the per-pixel accumulators (N > 1), rank 0 resolves
(gamma + clamp + 8-bit pack, src/vectors.h:372-384, src/scene.c:76-82) and copies the 8-bit image to the host.
The scene (flattened, device layout) and the per-rank pixel positions are resident in HBM before the timed region.

Workload (config.workload): wine_glass.acn at 1920x1080, path_samples 64, direct_samples 200 -- the scene and
sampling of BASELINE.json configs[1] at the resolution its metric is quoted on ("Msamples/s at 1920x1080").
`--workload c2` runs configs[1] verbatim (1280x720).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM as BASELINE.json asks (the path
is fp64-ALU bound, so frac is tiny by nature -- see DESIGN.md); `cpu_baseline` is the CPU oracle (a port of the
reference's algorithm; the reference itself needs the absent library beth) timed on the host cores on a strided
pixel subset of the same frame.
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
GRID_WORKGROUPS = 512    # persistent kernels of a lane: 2 workgroups of 256 lanes per compute unit (create_lane in actinon_hip.hip)
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


def cpu_baseline(flat, width, height, path_samples, target_seconds=15.0):
    """CPU oracle on a strided pixel subset of the same frame (same scene, same sampling), all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_binding import Oracle
    import actinon_amd as A
    o = Oracle()
    cores = host_core_share()
    pos = A.main_pass_positions(width, height)
    n = pos.shape[0]
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
    unit_scale = max(path_samples, 1)
    # the reference algorithm's fp64 work per pixel (cost table of SURVEY.md App. B), counted on the probe
    _, cnt = o.render_positions(flat, probe, linear=True, threads=cores, counters=True)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        model = "unknown"
    return {
        "value": sample.shape[0] * unit_scale / dt / 1e6,
        "unit": "Msamples/s" if path_samples else "Mpixels/s",
        "cores": cores,
        "kind": "port",
        "cpu_model": model,
        "flop_per_pixel": cnt["flop"] / probe.shape[0],
        "transcendentals_per_pixel": cnt["transc"] / probe.shape[0],
        "sample": f"every {stride}th pixel of the {width}x{height} frame ({sample.shape[0]} pixels x {path_samples} "
                  f"path samples, {dt:.1f} s, {cores} threads)",
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
    ap.add_argument("--save-image", default=None, help="write the last frame as PNM (rank 0)")
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
    n_rank = adist.rank_count(n_pix, rank, world)
    padded = adist.padded(n_pix, world)
    part = torch.empty((padded, 3), dtype=torch.float64, device=dev)
    gathered = torch.empty((world * padded, 3), dtype=torch.float64, device=dev) if world > 1 else None
    frame = torch.zeros((n_pix, 3), dtype=torch.float64, device=dev)
    rgb8 = torch.empty((n_pix, 3), dtype=torch.uint8, device=dev)
    host_img = torch.empty((n_pix, 3), dtype=torch.uint8).pin_memory()
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms = []

    def step(record):
        if world > 1:
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
                handle.render_main_pass_dev(0, n_pix, frame.data_ptr(), linear=True, stream=stream)
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

    # kernel duration of the dominant kernel: HIP events recorded by the library on the launch stream
    if args.quick:
        kernel_ms.append(handle.last_kernel_ms())   # of the last timed step
        stages = handle.last_stages()
        counters = None
    else:
        handle.stage_timing = True          # per-launch events only here: they cost ~0.7 % of a frame
        for _ in range(min(2, max(1, args.steps))):
            step(True)
        stages = handle.last_stages()
        handle.stage_timing = False
        # one extra, untimed pass through the instrumented kernels for the work counters
        handle.count_work = True
        step(False)
        torch.cuda.synchronize()
        counters = handle.last_counters()
        handle.count_work = False
    k_ms = float(np.mean(kernel_ms))

    if rank == 0:
        unit = max(S, 1)
        value = n_pix * unit * args.steps / elapsed / 1e6
        # algorithmic HBM bytes of one main pass on this rank: the radiance written (24 B per
        # position; positions are generated on the device) + one read of the flattened scene per workgroup that runs
        scene_bytes = flat.n_nodes * 288 + flat.c.n_elems * 8      # GNode 192 B + GMat 96 B per node, elems twice
        workgroups = int(stages["walk_launches"] + stages["shade_launches"] + stages["hard_launches"]) * GRID_WORKGROUPS
        alg_bytes = n_rank * 24 + workgroups * scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
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
                       "pixel_subset": (f"every {args.pixel_stride}th pixel of the {W}x{H} raster" if args.pixel_stride > 1 else "all"), "partition": f"pixel tiles of {adist.TILE}, round-robin over {world} rank(s) (acn_shard_tile_*)",
                       "exchange": f"RCCL all_gather of the ranks' parts ({padded * 24} B each), no reduction" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, world),
                         "kernel": "all kernels of one main pass (k_walk passes, k_shade x 4 size classes, k_hard_shadow, k_hard_path, "
                                   "k_shade_hits, k_finalize), HIP events on the launch stream around the pass; with concurrent "
                                   "lanes the per-family sums below overlap and exceed it",
                         "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes,
                         "family_ms_summed_over_lanes": {"k_walk+k_shade_hits": stages["walk_ms"], "k_shade": stages["shade_ms"],
                                                         "k_hard_*": stages["hard_ms"], "k_finalize": stages["finalize_ms"]},
                         "traffic_profile": "profiles/traffic.json (stamped with the kernel source hash; null when stale)",
                         "note": "the path is fp64-VALU-issue bound with divergent CSG traversal; the HBM roofline is "
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
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, W, H, S)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
        if args.save_image:
            from actinon_amd._lib import host
            img = host_img.numpy().astype(np.float64) / 256.0 + 0.5 / 256.0
            host.acn_write_pnm(args.save_image.encode(), img.ctypes.data, W, H)

    handle.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
