import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import actinon_amd as A
name = sys.argv[1] if len(sys.argv) > 1 else 'hanging_lamp'
ov = dict(image_width=60, image_height=80, direct_samples=8, path_samples=8)
flat = A.Scene.build(name, **ov).flatten() if name in ('diamond', 'wine_glass', 'primitives') else A.Flat.load(os.path.join(ROOT, 'tests/golden/scenes', name + '.npz'), **ov)
pos = A.main_pass_positions(60, 80)
imgs = {}
for v in ('1000000', '32'):
    os.environ['ACN_PRUNE_MIN'] = v
    h = A.Handle(flat, count_work=True)
    imgs[v] = h.render_positions(pos, linear=True)
    print(v, h.last_counters(), {k: h.last_stages()[k] for k in ('hard_rays',)})
    h.close()
d = np.abs(imgs['32'] - imgs['1000000']).max(axis=1)
bad = np.nonzero(d > 1e-9)[0]
print('differing pixels', len(bad), 'max', d.max())
for i in bad[:20]:
    print(i % 60, i // 60, imgs['32'][i], imgs['1000000'][i])
