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
# progressive JPEG (SOF2): Pillow writes libjpeg's standard scan script -- an interleaved DC scan, AC bands per component,
# then refinement scans of both kinds -- so all four coding procedures of T.81 annex G are exercised
for name, kw, grey, size in [("jpg_prog_444_q90.jpg", dict(quality=90, subsampling=0, progressive=True), False, (45, 31)),
                             ("jpg_prog_420_q75.jpg", dict(quality=75, subsampling=2, progressive=True), False, (50, 37)),
                             ("jpg_prog_422_q50_opt.jpg", dict(quality=50, subsampling=1, progressive=True, optimize=True), False, (64, 48)),
                             ("jpg_prog_grey_q85.jpg", dict(quality=85, progressive=True), True, (33, 17)),
                             ("jpg_prog_420_restart.jpg", dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=2), False, (70, 41)),
                             ("jpg_prog_1x1.jpg", dict(quality=90, subsampling=2, progressive=True), False, (1, 1)),
                             ("jpg_prog_420_q95_large.jpg", dict(quality=95, subsampling=2, progressive=True), False, (203, 131))]:
    im = Image.fromarray(picture(*size))
    if grey:
        im = im.convert("L")
    im.save(os.path.join(HERE, name), **kw)
    assert b"\xff\xc2" in open(os.path.join(HERE, name), "rb").read()
    expected[name] = np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"))[..., ::-1]
# refused on purpose: a progressive file cut before its last scans (libjpeg would show it smoothed)
whole = open(os.path.join(HERE, "jpg_prog_420_q75.jpg"), "rb").read()
cut = whole.rfind(b"\xff\xda")
open(os.path.join(HERE, "refused_progressive_cut.jpg"), "wb").write(whole[:cut])
if os.path.exists(os.path.join(HERE, "refused_progressive.jpg")):
    os.remove(os.path.join(HERE, "refused_progressive.jpg"))


# interlaced PNG (Adam7): Pillow cannot write it, so the files are assembled here (zlib + the seven reduced images, each row
# with one of the five filter types); what they must decode to is what Pillow (libpng) reads back
def png_chunk(kind, body):
    import struct
    import zlib
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def pack_row(samples, depth):
    """samples [n] of one row (channels interleaved) -> bytes at `depth` bits per sample"""
    if depth == 8:
        return bytes(samples.astype(np.uint8))
    if depth == 16:
        return samples.astype(">u2").tobytes()
    per = 8 // depth
    pad = (-len(samples)) % per
    v = np.concatenate([samples, np.zeros(pad, samples.dtype)]).reshape(-1, per).astype(np.uint32)
    shifts = np.array([(per - 1 - i) * depth for i in range(per)], np.uint32)
    return bytes((v << shifts).sum(1).astype(np.uint8))


def filter_row(ft, cur, up, bpp):
    cur = np.frombuffer(cur, np.uint8).astype(np.int32)
    up = np.frombuffer(up, np.uint8).astype(np.int32) if up is not None else np.zeros_like(cur)
    out = np.zeros_like(cur)
    for i in range(len(cur)):
        a = cur[i - bpp] if i >= bpp else 0
        b = up[i]
        c = up[i - bpp] if i >= bpp else 0
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) >> 1
        else:
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
        out[i] = (cur[i] - pred) & 255
    return bytes([ft]) + bytes(out.astype(np.uint8))


def write_adam7(path, samples, ctype, depth, palette=None):
    """samples [h][w][channels] (already at `depth` bits) -> interlaced PNG"""
    import struct
    import zlib
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    row_counter = 0
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for r in range(sub.shape[0]):
            cur = pack_row(sub[r].reshape(-1), depth)
            raw += filter_row(row_counter % 5, cur, prev, bpp)
            prev = cur
            row_counter += 1
    body = png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1))
    if palette is not None:
        body += png_chunk(b"PLTE", bytes(palette.astype(np.uint8).reshape(-1)))
    comp = zlib.compress(raw, 9)
    half = len(comp) // 2                                        # two IDAT chunks: the stream may be split anywhere
    body += png_chunk(b"IDAT", comp[:half]) + png_chunk(b"IDAT", comp[half:]) + png_chunk(b"IEND", b"")
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + body)


pic = picture(45, 31)
adam7_cases = [("png_adam7_RGB.png", pic.astype(np.uint16), 2, 8, None),
               ("png_adam7_RGBA16.png", np.concatenate([pic.astype(np.uint16) * 257, np.full((31, 45, 1), 40000, np.uint16)], -1), 6, 16, None),
               ("png_adam7_L1.png", (pic[..., :1] > 128).astype(np.uint16), 0, 1, None),
               ("png_adam7_L4.png", (pic[..., :1] >> 4).astype(np.uint16), 0, 4, None),
               ("png_adam7_P2.png", (pic[..., :1] >> 6).astype(np.uint16), 3, 2, np.array([[255, 0, 0], [0, 200, 30], [10, 20, 250], [90, 90, 90]])),
               ("png_adam7_LA.png", np.concatenate([pic[..., :1], 255 - pic[..., 1:2]], -1).astype(np.uint16), 4, 8, None),
               ("png_adam7_3x2.png", picture(3, 2).astype(np.uint16), 2, 8, None),       # passes 2..: some reduced images are empty
               ("png_adam7_1x1.png", picture(1, 1).astype(np.uint16), 2, 8, None)]
for name, samples, ctype, depth, palette in adam7_cases:
    write_adam7(os.path.join(HERE, name), samples, ctype, depth, palette)
    im = Image.open(os.path.join(HERE, name))
    assert im.info.get("interlace") == 1
    if im.mode in ("I;16", "I;16B", "I"):
        expected[name] = np.repeat((np.asarray(im) >> 8).astype(np.uint8)[..., None], 3, -1)
    elif name == "png_adam7_RGBA16.png":
        # Pillow reduces 16-bit RGBA to 8 bits per sample by keeping the high byte, as libpng's strip_16 (what cv::imread asks for) does
        expected[name] = np.asarray(im.convert("RGB"))[..., ::-1]
    else:
        expected[name] = np.asarray(im.convert("RGB"))[..., ::-1]
np.savez_compressed(os.path.join(HERE, "images_expected.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
print("bytes:", sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE)))
