"""Where the waves of the machine kernels spend their time: python scripts/phase_ticks.py [workload]
Needs a library built with EXTRA_DEFS=-DACN_PHASE_TIMERS (ACN_LIBDIR points at it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import actinon_amd as A
name = sys.argv[1] if len(sys.argv) > 1 else "wine_glass_1080p"
builder, ov = bench.WORKLOADS[name]
if builder.startswith("fixture:"):
    flat = A.Flat.load(os.path.join(bench.ROOT, "tests", "golden", "scenes", builder.split(":")[1] + ".npz"), **ov)
else:
    flat = A.Scene.build(builder, **ov).flatten()
w, h = int(flat.params.image_width), int(flat.params.image_height)
os.environ.setdefault("ACN_LANES", "1")
hd = A.Handle(flat)
import torch
out = torch.empty((w * h, 3), dtype=torch.float64, device="cuda:0")
for _ in range(2):
    hd.render_main_pass_dev(0, w * h, out.data_ptr(), linear=True)
torch.cuda.synchronize()
st = hd.last_stages()
print({k: round(v, 2) for k, v in st.items() if k.endswith("_ms")})
SHADE_NAMES = {"m_leaf": "cap sample", "m_pair": "light hit", "m_frame": "oren-nayar + intensity", "m_side": "occlusion test",
               "shade": "appends + sums", "compound": "path: sample + oren-nayar", "tail": "path: transition hit", "fetch": "task fetch + set-up"}
for kernel, ph in hd.last_phase_ticks().items():
    if kernel == "shade":     # slots 0 .. 3 hold the round tallies printed below
        ph = {SHADE_NAMES[p]: v for p, v in ph.items() if p in SHADE_NAMES}
    tot = sum(ph.values())  # the tally slots are outside the named phases
    if not tot:
        continue
    print(kernel, "total ticks %.3e" % tot)
    for p, v in ph.items():
        if v:
            print("   %-10s %5.1f %%" % (p, 100.0 * v / tot))
buf = hd.last_counters_raw(74)
for k, kernel in enumerate(["walk", "hard_shadow", "hard_path"]):
    t = [buf[10 + 16 * k + i] for i in (12, 13, 14, 15)]
    if t[1]:
        print("%s: %.1f of 64 lanes enter a lock-step machine (%d entries); %.1f lanes per leaf / in-line pair evaluation (%d)" % (kernel, t[0] / t[1], t[1], t[2] / max(t[3], 1), t[3]))
t = [buf[10 + 16 * 3 + i] for i in range(4)]
if t[1]:
    print("k_shade: %.1f of 64 lanes per round of the direct-light loops (%d rounds), %.1f per round of the path loops (%d)" % (t[0] / t[1], t[1], t[2] / max(t[3], 1), t[3]))
hd.close()
