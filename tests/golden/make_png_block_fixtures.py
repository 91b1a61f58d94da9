#!/usr/bin/env python3
"""Make statistical fixtures from the rendered PNGs the reference ships (run in the build
container, where /root/reference exists; the GPU box only sees the JSON this writes).

The reference's RNG is unseeded `rand::thread_rng()` (utils/random.rs:15-18), so its frames
cannot be reproduced sample for sample; what they pin is the *expectation* of the path.
For each PNG whose render settings are recorded in the reference source we store block
means of the linearised u8 image plus a mask of blocks that contain clamped pixels:

  output/output.png             Cornell box, src/main.rs:5-21 (300x300, spp 300, depth 20, bg 0.001)
  raytracer/output/quad_test.png    hittable/quad.rs:98-150 test_rendering (400x300, spp 10, depth 10)
  raytracer/output/render_test.png  renderer/renderer.rs:125-150 test_rendering (400x300, spp 3, depth 10)

Only numbers derived from pixels are written: no reference source text.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_png_blocks.json")
GAMMA = 2.2


def linearise(u8):
    # Image: c -> c^(1/2.2) -> clamp[0,0.999]*255 -> truncate (utils/image.rs:92-111); invert at bin centre
    return ((u8.astype(np.float64) + 0.5) / 255.0) ** GAMMA


def blocks(path, block):
    im = np.asarray(Image.open(os.path.join(REF, path)).convert("RGB"))
    h, w, _ = im.shape
    lin = linearise(im)
    bh, bw = h // block, w // block
    lin = lin[: bh * block, : bw * block].reshape(bh, block, bw, block, 3)
    mean = lin.mean(axis=(1, 3))
    sat = (im[: bh * block, : bw * block].reshape(bh, block, bw, block, 3) >= 254).any(axis=(1, 3, 4))
    return {"png": path, "width": w, "height": h, "block": block,
            "mean": np.round(mean, 6).tolist(), "saturated": sat.astype(int).tolist()}


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    data = {
        "_about": "block means of linearised reference PNGs; made by tests/golden/make_png_block_fixtures.py",
        "cornell": blocks("output/output.png", 20),
        "quad_test": blocks("raytracer/output/quad_test.png", 20),
        "render_test": blocks("raytracer/output/render_test.png", 20),
    }
    with open(OUT, "w") as f:
        json.dump(data, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
