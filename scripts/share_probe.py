"""What ONE rank of `world` does per frame under both splits of bench.py --gpus N, measured alone on one GPU (wine_glass 1080p p64 d200):
tiles (its 1/world of the pixel tiles, all samples) and samples (every pixel, its 1/world of the outermost sample loops).  The collective
(all_gather / all_reduce of the 50 MB frame) is not part of it.   usage: python scripts/share_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
import actinon_amd as A
from actinon_amd import dist as adist
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
builder, ov = bench.WORKLOADS["wine_glass_1080p"]
flat = A.Scene.build(builder, **ov).flatten()
n_pix = int(flat.params.image_width) * int(flat.params.image_height)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps
for world in (1, 2, 4, 8):
    h = A.Handle(flat, device=0)
    part = torch.empty((adist.padded(n_pix, world), 3), dtype=torch.float64, device=dev)
    t_tiles = timed(lambda: h.render_main_pass_shard_dev(0, n_pix, 0, world, part.data_ptr(), linear=True, stream=stream))
    h.close()
    h = A.Handle(flat, device=0)
    h.sample_shard = (0, world)
    frame = torch.zeros((n_pix, 3), dtype=torch.float64, device=dev)
    t_samples = timed(lambda: h.render_main_pass_dev(0, n_pix, frame.data_ptr(), linear=True, stream=stream))
    h.close()
    print("world %d: rank 0's share per frame  tiles %.2f ms   samples %.2f ms" % (world, t_tiles, t_samples), flush=True)
