"""One scene of tests/test_gpu_fuzz.py::test_random_sphere_field again: image difference between the GPU
frame and the double-precision restatement's, by depth limit (a chaotic scene parts from double
precision after some specular bounces; up to depth limit 2 no pixel may differ).
usage: python tests/replay/field_case.py SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

seed = int(sys.argv[1])
from oracle.pyoracle import Oracle                # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402
import test_gpu_fuzz as F                         # noqa: E402

oracle = Oracle()
rng = np.random.default_rng(50000 + seed)
flat, cam, scale, offset, side = F._random_field(S, rng, plain=seed % 3 == 1)
cd = oracle.camera(**cam)
W, H, spp = 60, 40, 4
ctx = api.Context(0)
ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH)
ctx.set_camera(**cam)
print("spheres", flat.n_prims, "scale", scale, "camera", cam["lookfrom"], "vfov", cam["vfov"], "depth of tree", ctx.stats()["bvh_depth"])
for depth in (1, 2, 3, 4, 6, 8, 12, 20, 50):
    sums, n_rays = oracle.render(flat, cd, W, H, spp, depth, 9 + seed, threads=8)
    img = ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed))
    plain = ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, traversal_slice=-1))
    st = ctx.stats()
    d = np.abs(img[..., :3] / spp - sums / spp)
    peak = max(1.0, float(sums.max() / spp))
    print(f"depth {depth:3d}: rmse {np.sqrt(np.mean(d ** 2)):.5f} (peak {peak:.2f})  pixels off by > 1e-4: {(d.max(axis=2) > 1e-4).mean():.4f}"
          f"  rays gpu {st['rays']} oracle {n_rays}  default == whole walk: {np.array_equal(img, plain)}", flush=True)
ctx.close()
