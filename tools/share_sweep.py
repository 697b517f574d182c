"""The headline frame by the share of unfinished walks at which a wave leaves the tree (traversal_exit_share, in
128ths of the lanes that came in) and by lanes per pixel.
usage: python tools/share_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

A = float(np.float32(16 / 9))
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.rtiow_camera(A).desc())
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")


def ms(**kw):
    best = 1e9
    for _ in range(4):
        ctx.render(api.Context.params(1920, 1080, 100, 50, 1, **kw), buf.data_ptr())
        best = min(best, ctx.stats()["kernel_ms"])
    return best


ms()
print("default", f"{ms():.3f} ms")
for share in (8, 16, 24, 32, 40, 48, 64):
    print(f"exit share {share:3d}/128: {ms(traversal_exit_share=share):.3f} ms", flush=True)
for k in (4, 8, 16):
    print(f"lanes per pixel {k:2d}: {ms(lanes_per_pixel=k):.3f} ms", flush=True)
ctx.close()
