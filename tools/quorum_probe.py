"""Sweep of the leaf quorum (RT_TUNE_QUORUM): how many lanes must stand on a leaf before a wave
stops walking inner nodes and tests leaves.  64 = wait for every lane."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
def bench(label, ctx, W, H, spp, values):
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ref = None
    for q in values:
        os.environ["RT_TUNE_QUORUM"] = str(q)
        ms = []
        for _ in range(2):
            ctx.render(api.Context.params(W, H, spp, 50, 1), buf.data_ptr()); st = ctx.stats(); ms.append(st["kernel_ms"])
        img = buf.cpu().numpy()
        if ref is None: ref = img
        print(f"{label} quorum {q:3d}: {min(ms):8.2f} ms  {st['rays'] / min(ms) / 1e3:9.1f} Mrays/s  same image: {np.array_equal(img, ref)}", flush=True)
aspect = float(np.float32(16 / 9))
c = api.Context(0)
vals = [0, 1, 2, 4, 8, 16, 32, 64]
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
bench("C2", c, 1920, 1080, 50, vals)
c.upload_scene(S.stress_flat(1000000), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
bench("C4", c, 3840, 2160, 8, vals)
