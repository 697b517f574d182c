"""Kernel time of one rank's shard (interleaved 4-row tiles) against lanes_per_pixel: the strong-scaling tail."""
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
def t(world, k, rank=0):
    ms = []
    for _ in range(3):
        c.render(api.Context.params(1920, 1080, 100, 50, 1, tiles=(4, rank, world) if world > 1 else (0, 0, 0), lanes_per_pixel=k), buf.data_ptr())
        ms.append(c.stats()["kernel_ms"])
    return min(ms)
full = t(1, 0)
print(f"full frame: {full:.2f} ms")
for world in (2, 4, 8):
    row = [f"{k}:{t(world, k):.2f}" for k in (0, 4, 8, 16, 32)]
    worst = max(t(world, 0, r) for r in range(world))
    print(f"world {world}: ideal {full/world:.2f} ms; K->ms " + "  ".join(row) + f"; slowest rank at K=auto {worst:.2f}", flush=True)
