"""TEST INFRASTRUCTURE — writes tests/golden/jpeg_cases.npz and tests/golden/reference_textures.json.

Inputs: small JPEG files made here with Pillow (synthetic pictures; every coding mode the
decoder handles), plus header-patched variants that reach the 1x2 and 4x1 upsamplers Pillow
cannot emit.  Expected outputs: the RGB8 bytes that the reference's own vendored decoder
(cuda/stb_image.h compiled by oracle/Makefile into oracle/_ref/libref_stb.so) produces for
them.  For the reference's texture files themselves (too large to commit; they stay in the
reference checkout) only SHA-256 digests of their decoded RGB8 are recorded.

Run from the repository root:  python oracle/gen_golden_jpeg.py
"""
import ctypes as C
import glob
import hashlib
import io
import json
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REFERENCE_DIR", "/root/reference")

R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_stb.so"))
R.ref_stbi_load_rgb.restype = C.c_void_p
R.ref_stbi_load_rgb.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
R.ref_stbi_free.argtypes = [C.c_void_p]


def stb_decode(data: bytes):
    w, h = C.c_int(), C.c_int()
    p = R.ref_stbi_load_rgb(data, len(data), C.byref(w), C.byref(h))
    if not p:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 3)).copy()
    R.ref_stbi_free(p)
    return out


def picture(w, h, seed):
    """Smooth gradients + hard edges + noise: exercises DC prediction, long zero runs and clamping."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 120 * np.sin(x / 5.0 + seed), 128 + 120 * np.cos(y / 3.0), (x * 7 + y * 13) % 256], -1)
    img[h // 3:h // 2, w // 4:w // 2] = (255, 0, 255)
    img += rng.normal(0, 12, img.shape)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")


def encode(img, **kw):
    b = io.BytesIO()
    img.save(b, "JPEG", **kw)
    return b.getvalue()


def patch_sampling(data: bytes, new_first, new_w, new_h):
    """Rewrites the luma sampling factors and the frame size in the SOF segment.  The entropy data
    is untouched: the number of blocks per MCU and of MCUs stays the same, so it still parses."""
    d = bytearray(data)
    i = 2
    while i < len(d):
        assert d[i] == 0xFF
        m = d[i + 1]
        L = (d[i + 2] << 8) | d[i + 3]
        if m in (0xC0, 0xC2):
            d[i + 5:i + 7] = new_h.to_bytes(2, "big")
            d[i + 7:i + 9] = new_w.to_bytes(2, "big")
            d[i + 11] = new_first
            return bytes(d)
        i += 2 + L
    raise ValueError("no SOF")


cases = {}
sizes = [(48, 40), (37, 29), (8, 8), (1, 1), (17, 9), (64, 3), (5, 70)]
for k, (w, h) in enumerate(sizes):
    img = picture(w, h, k)
    cases[f"base444_{w}x{h}"] = encode(img, quality=90, subsampling=0)
    cases[f"base420_{w}x{h}"] = encode(img, quality=75, subsampling=2)
    cases[f"base422_{w}x{h}"] = encode(img, quality=60, subsampling=1)
    cases[f"prog420_{w}x{h}"] = encode(img, quality=85, subsampling=2, progressive=True)
    cases[f"prog444_{w}x{h}"] = encode(img, quality=95, subsampling=0, progressive=True, optimize=True)
img = picture(96, 80, 11)
cases["grey_96x80"] = encode(img.convert("L"), quality=80)
cases["grey_prog_96x80"] = encode(img.convert("L"), quality=80, progressive=True)
cases["opt_huff_96x80"] = encode(img, quality=30, subsampling=2, optimize=True)
cases["q100_96x80"] = encode(img, quality=100, subsampling=0)
cases["q5_96x80"] = encode(img, quality=5, subsampling=2)
cases["restart_rows_96x80"] = encode(img, quality=80, subsampling=2, restart_marker_rows=1)
cases["restart_blocks_96x80"] = encode(img, quality=80, subsampling=0, restart_marker_blocks=3)
cases["restart_prog_96x80"] = encode(img, quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)
cases["cmyk_48x40"] = encode(picture(48, 40, 5).convert("CMYK"), quality=85)
# 2x1 luma (4:2:2) re-labelled 1x2: 32x16 -> 16x32 keeps 2x2 MCUs; reaches the vertical-only upsampler
cases["patched_h1v2_16x32"] = patch_sampling(encode(picture(32, 16, 3), quality=85, subsampling=1), 0x12, 16, 32)
# 2x2 luma (4:2:0) re-labelled 4x1: 32x32 -> 64x16 keeps 2x2 MCUs; reaches the nearest-neighbour upsampler
cases["patched_h4v1_64x16"] = patch_sampling(encode(picture(32, 32, 4), quality=85, subsampling=2), 0x41, 64, 16)
cases["patched_h1v4_16x64"] = patch_sampling(encode(picture(32, 32, 6), quality=85, subsampling=2), 0x14, 16, 64)

out = {}
for name, data in cases.items():
    rgb = stb_decode(data)
    assert rgb is not None, name
    out[name + "/jpeg"] = np.frombuffer(data, np.uint8)
    out[name + "/rgb"] = rgb
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"), **out)
print(len(cases), "cases,", sum(len(v) for v in cases.values()), "JPEG bytes")

digests = {}
for f in sorted(glob.glob(os.path.join(REF, "cuda", "textures", "**", "*.jpg"), recursive=True)):
    rgb = stb_decode(open(f, "rb").read())
    rel = os.path.relpath(f, os.path.join(REF, "cuda"))
    digests[rel] = dict(width=int(rgb.shape[1]), height=int(rgb.shape[0]), sha256_rgb8=hashlib.sha256(rgb.tobytes()).hexdigest())
json.dump(digests, open(os.path.join(ROOT, "tests", "golden", "reference_textures.json"), "w"), indent=1)
print(len(digests), "reference textures digested")
