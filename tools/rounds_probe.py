"""Frame time of the plain walk (traversal_slice = -1) against the walk left between rounds of leaves by the
exit share (traversal_slice = 0 on a tree of at most 16 levels, traversal_exit_share = share).
usage: python tools/rounds_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402


def sweep(name, flat, cam, W, H, spp, shares, flags=0):
    ctx = api.Context(0)
    ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH)
    ctx.set_camera(**cam)
    buf = torch.zeros((H, W, 4), device="cuda")
    ref = None
    print(f"{name}: depth {ctx.stats()['bvh_depth']}, {W}x{H}@{spp}", flush=True)
    for share in shares:
        best, rays = None, 0
        for _ in range(3):
            ctx.render(api.Context.params(W, H, spp, 50, 1, flags=flags, traversal_slice=-1 if share is None else 0,
                                          traversal_exit_share=-1 if share is None else share), buf.data_ptr())
            st = ctx.stats()
            best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
        img = buf.cpu().numpy()
        if ref is None:
            ref = img
        print(f"  share {str(share):>5}: {best:8.3f} ms   same image {np.array_equal(img, ref)}", flush=True)
    ctx.close()


A = 16 / 9
shares = (None, 16, 32, 48, 64, 80, 96, 112)
sweep("rtiow 487 spheres (tree in LDS)", S.flatten(S.rtiow_scene()), S.rtiow_camera(A).desc(), 1920, 1080, 100, shares)
sweep("rtiow 487 spheres (tree in HBM)", S.flatten(S.rtiow_scene()), S.rtiow_camera(A).desc(), 1920, 1080, 100, shares, flags=api.RT_FLAG_NO_LDS_SCENE)
sweep("skybox world", S.flatten(S.skybox_scene(), background=S.RT_BG_SOLID), S.skybox_camera(A).desc(), 1920, 1080, 100, (None, 32, 64, 96))
scam = S.stress_camera(A).desc()
for n in (20_000, 50_000):
    sweep(f"{n} spheres", S.stress_flat(n), scam, 1920, 1080, 64, (None, 32, 48, 64, 80, 96))
