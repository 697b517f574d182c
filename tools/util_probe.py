import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for depth in (1, 2, 3, 50):
    for k in (1, 4, 16):
        c.render(api.Context.params(1920, 1080, 20, depth, 1, flags=api.RT_FLAG_COUNTERS, lanes_per_pixel=k), buf.data_ptr())
        st = c.stats()
        print(f"depth {depth:2d} K={k:2d}: rays {st['rays']:.3e} steps/ray {st['phase_clocks'][0] / st['rays']:.2f} wave steps/trip {st['phase_clocks'][1] / (st['rays'] / 64):.2f} util {st['phase_clocks'][0] / (64.0 * st['phase_clocks'][1]):.3f} ms {st['kernel_ms']:.2f}")
