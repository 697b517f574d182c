"""How much of a frame is the drain at the end of each wave tile (the last paths of a tile run on a
few lanes)?  More samples per pixel amortise it: Mrays/s by spp."""
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
for K in (2, 4, 8, 16, 32):
    for spp in (25, 100, 1600):
        ms = []
        for _ in range(3):
            c.render(api.Context.params(1920, 1080, spp, 50, 1, lanes_per_pixel=K), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
        print(f"K={K} spp={spp} nodes={st['bvh_nodes']} depth={st['bvh_depth']}: {min(ms):8.2f} ms  {st['rays'] / min(ms) / 1e3:8.1f} Mrays/s", flush=True)
