import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
c.upload_scene(S.stress_flat(int(os.environ.get("C4_SPHERES", "1000000"))), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
for flags, name in ((0, "plain"), (api.RT_FLAG_PHASE_CLOCKS, "phase clocks")):
    c.render(api.Context.params(3840, 2160, 8, 50, 1, flags=flags), buf.data_ptr()); st = c.stats()
    print(name, st["kernel_ms"], "ms", st["rays"] / st["kernel_ms"] / 1e3, "Mrays/s")
pc = np.array(st["phase_clocks"], dtype=np.float64)
print("phase share: camera %.3f always %.3f traversal %.3f record %.3f scatter %.3f" % tuple(pc / pc.sum()))
