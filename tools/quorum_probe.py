"""Frame time of the headline configuration walked by the time-sliced kernel with a slice too long to
matter, by leaf quorum (rt_render_params.leaf_quorum): how early a wave turns to its leaves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytrace_amd import api, scene as S
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.rtiow_camera(float(np.float32(16 / 9))).desc())
buf = torch.zeros((1080, 1920, 4), device="cuda")
for sl, q in [(0, 0)] + [(100000, q) for q in (-1, 1, 2, 3, 4, 6, 8, 16, 64)] + [(16, 2), (32, 2), (64, 2), (32, 4), (64, 4)]:
    best = None
    for _ in range(3):
        ctx.render(api.Context.params(1920, 1080, 100, 50, 1, traversal_slice=sl, leaf_quorum=q), buf.data_ptr())
        st = ctx.stats()
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    ctx.render(api.Context.params(1920, 1080, 100, 50, 1, traversal_slice=sl, leaf_quorum=q, flags=api.RT_FLAG_COUNTERS), buf.data_ptr())
    c = ctx.stats()
    print(f"slice {sl:6d} quorum {q:3d}: {best:7.3f} ms  box tests/ray {c['box_tests'] / c['rays']:.2f}  "
          f"traversal lane util {c['phase_clocks'][0] / (64.0 * c['phase_clocks'][1]):.3f}", flush=True)
