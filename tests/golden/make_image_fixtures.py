#!/usr/bin/env python3
"""Generates tests/golden/images/*: small PNG / JPEG files written by Pillow (libpng / libjpeg-turbo) and
images_expected.npz, the pixels Pillow's own decoders return for them (B,G,R).  The product's decoders
(csrc/host/rt_image_io.cpp) must reproduce these bytes exactly: PNG is lossless, and the JPEG reader restates
libjpeg's default arithmetic (integer "islow" IDCT, fancy upsampling, 16-bit colour conversion), which is what
cv::imread -- the reference's texture decoder, Material.hpp:29-43 -- runs as well.
    python tests/golden/make_image_fixtures.py        (needs Pillow; run once, outputs are committed)"""
import os
import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(20240917)


def picture(w, h):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos((x + y) / 11.0), x * 255 // max(w - 1, 1) + 0 * y], -1)
    return np.clip(img + rng.normal(0, 14, img.shape), 0, 255).astype(np.uint8)


expected = {}
rgb = picture(45, 31)
for mode in ("RGB", "L", "RGBA", "P", "LA", "1"):
    name = "png_%s.png" % mode
    Image.fromarray(rgb).convert(mode).save(os.path.join(HERE, name), optimize=mode in ("RGB", "P"))
    expected[name] = np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"))[..., ::-1]
name = "png_I16.png"
Image.fromarray(rgb[..., 0].astype(np.uint16) * 257 + 3).save(os.path.join(HERE, name))
expected[name] = np.repeat((np.asarray(Image.open(os.path.join(HERE, name))) >> 8).astype(np.uint8)[..., None], 3, -1)
for name, kw, grey, size in [("jpg_444_q90.jpg", dict(quality=90, subsampling=0), False, (45, 31)),
                             ("jpg_422_q60.jpg", dict(quality=60, subsampling=1), False, (45, 31)),
                             ("jpg_420_q75.jpg", dict(quality=75, subsampling=2), False, (45, 31)),
                             ("jpg_420_q35_opt.jpg", dict(quality=35, subsampling=2, optimize=True), False, (64, 48)),
                             ("jpg_grey_q80.jpg", dict(quality=80), True, (33, 17)),
                             ("jpg_420_restart.jpg", dict(quality=85, subsampling=2, restart_marker_blocks=3), False, (50, 37)),
                             ("jpg_1x1.jpg", dict(quality=90, subsampling=2), False, (1, 1))]:
    im = Image.fromarray(picture(*size))
    if grey:
        im = im.convert("L")
    im.save(os.path.join(HERE, name), **kw)
    expected[name] = np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"))[..., ::-1]
# refused on purpose
Image.fromarray(picture(40, 30)).save(os.path.join(HERE, "refused_progressive.jpg"), progressive=True)
np.savez_compressed(os.path.join(HERE, "images_expected.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
print("bytes:", sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE)))
