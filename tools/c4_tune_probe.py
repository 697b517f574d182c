"""10^6-sphere scene, 4K at 16 spp: frame time by traversal time slice (RT_TUNE_SLICE) and leaf
quorum (RT_TUNE_QUORUM)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
c.upload_scene(S.stress_flat(1000000), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
for sl in (16, 24, 32, 48, 64):
    for q in (1, 2, 3, 4):
        os.environ["RT_TUNE_SLICE"] = str(sl); os.environ["RT_TUNE_QUORUM"] = str(q)
        ms = []
        for _ in range(2):
            c.render(api.Context.params(3840, 2160, 16, 50, 1), buf.data_ptr()); ms.append(c.stats()["kernel_ms"])
        print(f"slice {sl:3d} quorum {q}: {min(ms):7.2f} ms", flush=True)
