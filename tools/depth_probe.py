"""The headline frame by depth limit: time, rays, box tests per ray and traversal lane utilisation of the
camera rays alone (depth limit 1), of paths of at most 2, 3, 5 rays, and of the whole frame.  Tells how the
frame's time splits between camera rays (coherent within a 4x2 tile) and bounce rays (not).
usage: python tools/depth_probe.py"""
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
for _ in range(3):
    ctx.render(api.Context.params(1920, 1080, 100, 50, 1), buf.data_ptr())
    ctx.stats()
for depth in (1, 2, 3, 5, 50):
    ms = []
    for _ in range(3):
        ctx.render(api.Context.params(1920, 1080, 100, depth, 1), buf.data_ptr())
        st = ctx.stats()
        ms.append(st["kernel_ms"])
    ctx.render(api.Context.params(1920, 1080, 100, depth, 1, flags=api.RT_FLAG_COUNTERS), buf.data_ptr())
    c = ctx.stats()
    util = c["phase_clocks"][0] / (64.0 * c["phase_clocks"][1]) if c["phase_clocks"][1] else 0
    print(f"depth limit {depth:2d}: {min(ms):7.3f} ms  {c['rays'] / 1e6:7.1f} Mrays  {c['rays'] / min(ms) / 1e6:6.2f} Grays/s  "
          f"box tests/ray {c['box_tests'] / c['rays']:5.2f}  sphere tests/ray {c['sphere_tests'] / c['rays']:4.2f}  "
          f"lane steps/ray {c['phase_clocks'][0] / c['rays']:5.2f}  wave steps x64/ray {64 * c['phase_clocks'][1] / c['rays']:6.2f}  traversal lane utilisation {util:.3f}", flush=True)
ctx.close()
