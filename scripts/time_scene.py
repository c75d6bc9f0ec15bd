"""One render of a scene at given size/sampling on the GPU; prints time and pipeline stats (no oracle)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import actinon_amd as A
name, w, h, ps, ds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
t0 = time.time()
sc = A.Scene.build(name, image_width=w, image_height=h, path_samples=ps, direct_samples=ds)
flat = sc.flatten()
print("build+flatten %.2f s nodes %d" % (time.time() - t0, flat.n_nodes), flush=True)
H = A.Handle(flat)
H.stage_timing = True
out = torch.zeros((w * h, 3), dtype=torch.float64, device="cuda:0")
for it in range(2):
    t0 = time.time()
    H.render_main_pass_dev(0, w * h, out.data_ptr(), linear=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = H.last_stages()
    print("iter %d: %.3f s  %.1f Msamples/s  stages %s" % (it, dt, w * h * max(ps, 1) / dt / 1e6, {k: round(v, 1) for k, v in st.items()}), flush=True)
img = out.cpu().numpy()
print("mean rgb", img.mean(axis=0), "finite", np.isfinite(img).all())
if len(sys.argv) > 6:
    from PIL import Image
    Image.fromarray(A.cps_from_cl(img).reshape(h, w, 3)).save(sys.argv[6])
