import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
for flags, name in ((0, "plain"), (0, "plain"), (api.RT_FLAG_PHASE_CLOCKS, "phase clocks"), (api.RT_FLAG_COUNTERS, "counters")):
    c.render(api.Context.params(1920, 1080, 100, 50, 1, flags=flags), buf.data_ptr()); st = c.stats()
    print(name, round(st["kernel_ms"], 2), "ms", round(st["rays"] / st["kernel_ms"] / 1e3), "Mrays/s")
    if flags == api.RT_FLAG_PHASE_CLOCKS:
        pc = np.array(st["phase_clocks"], dtype=np.float64)
        print("phase share: camera %.3f always %.3f traversal %.3f record %.3f scatter+shade %.3f" % tuple(pc / pc.sum()))
    if flags == api.RT_FLAG_COUNTERS:
        r = st["rays"]
        print("per ray: box %.2f sph %.2f hits %.2f wide %.2f; steps/ray %.2f, lane util %.3f" % (st["box_tests"] / r, st["sphere_tests"] / r, st["sphere_hits"] / r, st["wide_tests"] / r, st["phase_clocks"][0] / r, st["phase_clocks"][0] / max(1, st["phase_clocks"][1]) / 64))
