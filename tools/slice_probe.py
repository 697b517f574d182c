"""Frame time of the stress scene (tree walked out of HBM) by traversal slice and leaf quorum
(rt_render_params.traversal_slice, leaf_quorum).  usage: python tools/slice_probe.py [spp] [n_spheres] [width height]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
ctx = api.Context(0)
ctx.upload_scene(S.stress_flat(n), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.stress_camera(16 / 9).desc())
buf = torch.zeros((H, W, 4), device="cuda")
print(f"{n} spheres, depth {ctx.stats()['bvh_depth']}, {W}x{H}@{spp}", flush=True)
cases = [(0, 0), (-1, 0)] + [(sl, q) for sl in (32, 48, 64, 96, 128, 192, 256) for q in (2, 4, 6)]
for sl, q in cases:
    best = None
    for _ in range(2):
        ctx.render(api.Context.params(W, H, spp, 50, 1, traversal_slice=sl, leaf_quorum=q), buf.data_ptr())
        ms = ctx.stats()["kernel_ms"]
        best = ms if best is None else min(best, ms)
    print(f"slice {sl:4d} quorum {q:3d}: {best:8.3f} ms", flush=True)
