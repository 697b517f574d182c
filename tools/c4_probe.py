import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
def run(label, W, H, spp, **kw):
    ms = []
    for _ in range(3):
        c.render(api.Context.params(W, H, spp, 50, 1, **kw), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
    print(f"{label}: {min(ms):8.2f} ms  {st['rays'] / min(ms) / 1e3:9.1f} Mrays/s", flush=True)
run("C2 1080p@100", 1920, 1080, 100)
n = int(os.environ.get("C4_SPHERES", "1000000"))
c.upload_scene(S.stress_flat(n), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
run("C4 4K@16", 3840, 2160, 16)
c.render(api.Context.params(3840, 2160, 4, 50, 1, flags=api.RT_FLAG_COUNTERS), buf.data_ptr()); st = c.stats()
print("C4 trav util", st["phase_clocks"][0] / (64.0 * st["phase_clocks"][1]), {k: round(st[k] / st["rays"], 2) for k in ("box_tests", "sphere_tests")})
