"""Strong-scaling probe on ONE GPU: renders the share of pixels one rank of `world` would get (wine_glass 1080p),
with K concurrently driven handles (own stream + workspace each, one host thread per handle).
usage: scale_probe.py <world> <K> [steps]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import actinon_amd as A
from actinon_amd import dist as adist
from concurrent.futures import ThreadPoolExecutor

world, K = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else K          # chunks; K worker handles pull them from a queue
mode = sys.argv[4] if len(sys.argv) > 4 else "interleave"  # or "contiguous"
steps = 5
flat = A.Scene.build("wine_glass", image_width=1920, image_height=1080, path_samples=64, direct_samples=200).flatten()
W, H = 1920, 1080
idx = adist.rank_pixels(W * H, 0, world)
dev = torch.device("cuda:0")
parts = np.array_split(np.arange(idx.shape[0]), K)      # contiguous runs of this rank's tiles
# interleave tiles over the K handles instead (balances glass / floor): tile t of the rank goes to handle t % K
tile = np.arange(idx.shape[0]) // adist.TILE
if mode == "interleave":
    parts = [np.nonzero(tile % M == m)[0] for m in range(M)]
else:
    parts = np.array_split(np.arange(idx.shape[0]), M)
handles, poss, outs, streams = [], [], [], []
for k in range(K):
    handles.append(A.Handle(flat, device=0))
    streams.append(torch.cuda.Stream())
for m in range(M):
    p = torch.from_numpy(adist.pixel_positions(idx[parts[m]], W)).to(dev)
    poss.append(p); outs.append(torch.empty((p.shape[0], 3), dtype=torch.float64, device=dev))
pool = ThreadPoolExecutor(max_workers=K)
import itertools
lock = threading.Lock()

def worker(k, counter):
    while True:
        with lock:
            m = next(counter, None)
        if m is None:
            return
        handles[k].render_positions_dev(poss[m].data_ptr(), poss[m].shape[0], outs[m].data_ptr(), linear=True, stream=streams[k].cuda_stream)
        streams[k].synchronize()

def step():
    counter = iter(range(M))
    list(pool.map(lambda k: worker(k, counter), range(K)))

for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("world %d  K %d  M %d  %s  pixels %d  %.2f ms" % (world, K, M, mode, idx.shape[0], dt * 1e3), flush=True)
