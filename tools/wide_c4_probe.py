"""10^6-sphere frame (4K, 16 spp): binary against four-wide tree by traversal time slice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
flat, cam = S.stress_flat(1_000_000), S.stress_camera(aspect).desc()
buf = torch.zeros((2160, 3840, 4), device="cuda")
for label, accel, slices in (("bvh2", api.RT_ACCEL_BVH, (0, 32, 64)), ("bvh4", api.RT_ACCEL_BVH4, (8, 12, 16, 24, 32, 48))):
    ctx = api.Context(0)
    ctx.upload_scene(flat, accel=accel)
    ctx.set_camera(**cam)
    for sl in slices:
        for q in (0, 1, 4):
            best = None
            for _ in range(2):
                ctx.render(api.Context.params(3840, 2160, 16, 50, 1, traversal_slice=sl, leaf_quorum=q), buf.data_ptr())
                st = ctx.stats()
                best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
            print(f"c4 {label} slice {sl:3d} quorum {q}: {best:8.3f} ms", flush=True)
    ctx.close()
