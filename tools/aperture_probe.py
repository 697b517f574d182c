import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
c = api.Context(0)
c.upload_scene(S.flatten(S.rtiow_scene()))
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for ap in (0.1, 0.0):
    c.set_camera((6, 3, -13), (0, 0, 0), (0, 1, 0), 20.0, float(np.float32(16 / 9)), ap, 10.0)
    ms = []
    for _ in range(3):
        c.render(api.Context.params(1920, 1080, 100, 50, 1), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
    print("aperture", ap, round(min(ms), 2), "ms", st["rays"])
