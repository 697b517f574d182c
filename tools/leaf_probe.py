"""Frame time of slabs of 2 x 10^4, 10^5 and 10^6 random spheres (1920x1080 @16, depth 50) on the host-built
tree, with tests per ray: for comparing leaf-size rules of the builder.  usage: python tools/leaf_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

W, H, spp = 1920, 1080, 16
buf = torch.zeros((H, W, 4), device="cuda")
for n in (20_000, 100_000, 1_000_000):
    ctx = api.Context(0)
    ctx.upload_scene(S.stress_flat(n), accel=api.RT_ACCEL_BVH)
    ctx.set_camera(**S.stress_camera(16 / 9).desc())
    best = None
    for _ in range(3):
        ctx.render(api.Context.params(W, H, spp, 50, 1), buf.data_ptr())
        ms = ctx.stats()["kernel_ms"]
        best = ms if best is None else min(best, ms)
    ctx.render(api.Context.params(W, H, spp, 50, 1, flags=api.RT_FLAG_COUNTERS), buf.data_ptr())
    st = ctx.stats()
    print(f"n {n:8d}: {best:8.3f} ms  nodes {st['bvh_nodes']} depth {st['bvh_depth']} coarse {st['bvh_coarse_nodes']}  "
          f"box {st['box_tests'] / st['rays']:.1f} sphere {st['sphere_tests'] / st['rays']:.1f} per ray  build {st['build_ms']:.0f} ms", flush=True)
    ctx.close()
