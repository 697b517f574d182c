import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
for n, W, H, spp in ((1000000, 3840, 2160, 64), (100000, 1920, 1080, 64), (20000, 1920, 1080, 64)):
    c.upload_scene(S.stress_flat(n), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
    for sl in (32, 48, 64, 128):
        os.environ["RT_TUNE_SLICE"] = str(sl)
        ms = []
        for _ in range(2):
            c.render(api.Context.params(W, H, spp, 50, 1), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
        print(f"n {n} depth {st['bvh_depth']} slice {sl}: {min(ms):7.2f} ms", flush=True)
