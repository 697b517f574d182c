import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for accel, name in ((api.RT_ACCEL_BVH, "bvh"), (api.RT_ACCEL_LIST, "list")):
    c = api.Context(0)
    c.upload_scene(S.flatten(S.rtiow_scene()), accel=accel); c.set_camera(**S.rtiow_camera(aspect).desc())
    ms = []
    for _ in range(3):
        c.render(api.Context.params(1920, 1080, 20, 50, 1), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
    tf = st["rays"] * (487 * 25 + 70) / (min(ms) * 1e-3) / 1e12
    print(f"{name}: {min(ms):.2f} ms @20spp {st['rays'] / min(ms) / 1e3:.0f} Mrays/s; brute-force-equivalent {tf:.1f} TFLOP/s ({tf / 157.3:.3f} of peak)")
    c.close()
