"""The 10^6-sphere frame (3840x2160 @64) by traversal tuning: slice, leaf quorum and exit share of the time-sliced
walk (the image does not depend on any of them).  usage: python tools/c4_sweep.py"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

W, H, SPP = 3840, 2160, 64
ctx = api.Context(0)
ctx.upload_scene(S.stress_flat(1_000_000), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.stress_camera(float(np.float32(W / H))).desc())
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")


def t(**kw):
    ms = []
    for _ in range(2):
        ctx.render(api.Context.params(W, H, SPP, 50, 1, **kw), buf.data_ptr())
        ms.append(ctx.stats()["kernel_ms"])
    return min(ms)


print(f"defaults: {t():.2f} ms", flush=True)
for sl, q, sh in itertools.product((48, 64, 96), (3, 4, 6), (48, 64, 80)):
    print(f"slice {sl:3d} quorum {q} share {sh:3d}: {t(traversal_slice=sl, leaf_quorum=q, traversal_exit_share=sh):.2f} ms", flush=True)
for k in (4, 16):
    print(f"lanes per pixel {k}: {t(lanes_per_pixel=k):.2f} ms", flush=True)
print(f"defaults again: {t():.2f} ms")
