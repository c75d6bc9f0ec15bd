"""Whose are the ~100 ms of the first stream acn_scene_upload makes in a cold process?  Time a stream the HOST makes first (torch.cuda.Stream),
then the upload: python scripts/first_stream.py [host_stream_first=1] [workload]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["ACN_DEBUG_CHUNKS"] = "1"
import bench
import actinon_amd as A
import torch
host_first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "c2"
builder, ov = bench.WORKLOADS[name]
flat = A.Scene.build(builder, **ov).flatten()
torch.cuda.synchronize()
if host_first:
    for k in range(2):
        t0 = time.perf_counter(); s = torch.cuda.Stream(); torch.cuda.synchronize(); print("host stream %d: %.1f ms" % (k + 1, (time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter(); x = torch.zeros(1 << 20, device="cuda:0"); torch.cuda.synchronize(); print("host allocation + fill: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
hd = A.Handle(flat)
print("host_stream_first=%d  upload: %.1f ms" % (host_first, (time.perf_counter() - t0) * 1e3))
hd.close()
