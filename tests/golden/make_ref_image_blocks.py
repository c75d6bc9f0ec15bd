"""Generates tests/golden/ref_image_blocks.json from the reference's own shipped renders.

Run in the build container only (reads /root/reference/image/*.png|jpg; that tree does not exist on the GPU box):
    python tests/golden/make_ref_image_blocks.py

The reference has no tests or golden vectors; its rendered example images are the only outputs of
the reference that exist.  They were produced at each script's own settings (incl. adaptive anti-aliasing and the
unknown beth LCG), so they pin the path statistically, not per pixel: we store BxB block means of the 8-bit
images (B = 16 pixels) as data, not the images themselves.  Images are CC-BY-SA 4.0 (reference README.md:132),
(c) Johannes B. Steffens; this derived table inherits that licence.
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference/image"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_image_blocks.json")
BLOCK = 16
IMAGES = {"primitives": "primitives.acn.png", "wine_glass": "wine_glass.acn.png", "diamond": "diamond.acn.png",
          "many_spheres": "many_spheres.acn.png", "pyramid": "pyramid.acn.png", "ruby_heart": "ruby_heart.acn.png",
          "caustic_of_caustic": "caustic_of_caustic.acn.png", "paraffin_lamp": "paraffin_lamp.acn.png",
          "paraffin_lamp_on_ledge": "paraffin_lamp_on_ledge.acn.png", "hanging_lamp": "hanging_lamp_acn.png",
          # frame 49 of src_acn/diamond_video.acn (400x300, the gem of diamond.acn through a second script) and the README's
          # title picture: src_acn/hanging_lamps_in_row rendered at 3200x1800 and published scaled to 640x360 as JPEG
          "diamond_video_049": "diamond_video.acn.image_000049.png", "hanging_lamps_in_row": "hanging_lamp02.acn.640_360.jpg"}


def block_means(img, b):
    h, w, _ = img.shape
    hb, wb = h // b, w // b
    v = img[:hb * b, :wb * b].reshape(hb, b, wb, b, 3).astype(np.float64)
    return v.mean(axis=(1, 3))


def main():
    out = {"block": BLOCK, "unit": "8-bit value (0..255)", "source": "johsteffens/actinon image/*.png|jpg (CC-BY-SA 4.0)",
           "images": {}}
    for name, fn in IMAGES.items():
        img = np.asarray(Image.open(os.path.join(REF, fn)).convert("RGB"))
        bm = block_means(img, BLOCK)
        out["images"][name] = {"file": fn, "width": int(img.shape[1]), "height": int(img.shape[0]),
                               "block_means": np.round(bm, 3).tolist()}
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    # 8 x 8 blocks (edge blocks included: 400 / 8 = 50 exactly) of the two pictures the tight pin uses
    # (tests/test_reference_images.py::test_oracle_with_gradient_cycles_matches_shipped_render)
    out8 = {"block": 8, "unit": out["unit"], "source": out["source"], "images": {}}
    for name in ("primitives", "wine_glass"):
        img = np.asarray(Image.open(os.path.join(REF, IMAGES[name])).convert("RGB"))
        out8["images"][name] = {"file": IMAGES[name], "width": int(img.shape[1]), "height": int(img.shape[0]),
                                "block_means": np.round(block_means(img, 8), 3).tolist()}
    with open(OUT.replace("blocks.json", "blocks8.json"), "w") as f:
        json.dump(out8, f)
    print("wrote", OUT.replace("blocks.json", "blocks8.json"))


if __name__ == "__main__":
    main()
