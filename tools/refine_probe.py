"""Headline frame with and without the double-precision polish of the winning hit (RT_FLAG_NO_REFINE)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytrace_amd import api, scene as S
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.rtiow_camera(float(np.float32(16 / 9))).desc())
buf = torch.zeros((1080, 1920, 4), device="cuda")
for flags in (0, api.RT_FLAG_NO_REFINE, api.RT_FLAG_PHASE_CLOCKS):
    best = None
    for _ in range(4):
        ctx.render(api.Context.params(1920, 1080, 100, 50, 1, flags=flags), buf.data_ptr())
        st = ctx.stats()
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    pc = st["phase_clocks"]
    print(f"flags {flags}: {best:7.3f} ms rays {st['rays']}", [round(x / max(sum(pc), 1), 3) for x in pc], flush=True)
