"""Generates tests/golden/oracle_images.npz: linear radiance images of the small parity configurations rendered by
the CPU oracle (deterministic-math build).  They pin the oracle against accidental change and give the GPU tests a
committed expectation.  Run:  python tests/golden/make_oracle_goldens.py   (needs `make oracle host hip`)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle_binding import Oracle  # noqa: E402
import scenes_util as S  # noqa: E402


def main():
    o = Oracle()
    out = {}
    for name in S.SMALL:
        sc, flat = S.build(name)
        img = o.render_positions(flat, S.positions(flat), linear=True)
        img = img.reshape(flat.params.image_height, flat.params.image_width, 3)
        sub = S.GOLDEN_STRIDE.get(name, 1)
        out[name] = np.ascontiguousarray(img[::sub, ::sub])
        print(name, out[name].shape, float(out[name].mean()))
    path = os.path.join(HERE, "oracle_images.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
