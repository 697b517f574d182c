"""Frame time by lanes_per_pixel (K) for the bench configurations."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
def run(label, W, H, spp):
    for K in (4, 8, 16):
        ms = []
        for _ in range(3):
            c.render(api.Context.params(W, H, spp, 50, 1, lanes_per_pixel=K), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
        print("%s K=%d: %.2f ms %.0f Mrays/s" % (label, K, min(ms), st["rays"] / min(ms) / 1e3), flush=True)
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
run("C2", 1920, 1080, 100)
c.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_LIST)
run("C2 list@20", 1920, 1080, 20)
c.upload_scene(S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID)); c.set_camera(**S.cornell_camera(aspect).desc())
run("C3", 1920, 1080, 100)
c.upload_scene(S.stress_flat(1000000), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
run("C4@16", 3840, 2160, 16)
run("C4@64", 3840, 2160, 64)
