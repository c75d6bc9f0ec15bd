"""Per-frame wall time and pipeline statistics of consecutive main passes: python scripts/frame_times.py <workload> [frames]
(run from the root of the tree whose library is to be measured: it imports ./bench.py and ./actinon_amd)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
import actinon_amd as A
import torch
name = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
builder, ov = bench.WORKLOADS[name]
if builder.startswith("fixture:"):
    flat = A.Flat.load(os.path.join(bench.ROOT, "tests", "golden", "scenes", builder.split(":")[1] + ".npz"), **ov)
else:
    flat = A.Scene.build(builder, **ov).flatten()
w, h = int(flat.params.image_width), int(flat.params.image_height)
torch.cuda.synchronize()
t_up = time.perf_counter()
hd = A.Handle(flat)
t_up = (time.perf_counter() - t_up) * 1e3
out = torch.empty((w * h, 3), dtype=torch.float64, device="cuda:0")
line = []
for f in range(frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hd.render_main_pass_dev(0, w * h, out.data_ptr(), linear=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = hd.last_stages()
    line.append("%.1f ms (chunks %d, retries %d, walk launches %d, ws %.2f GB, allocs %s)" % (dt, st["chunks"], st["retries"], st["walk_launches"], st.get("workspace_bytes", 0) / 1e9, st.get("workspace_allocs", "-")))
print(name, os.getcwd().split("/")[-1] or "repo", "   upload (acn_scene_upload, first HIP work of the library in this process): %.1f ms" % t_up)
for l in line:
    print("   ", l)
hd.close()
