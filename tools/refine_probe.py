"""Cost and effect of the double-precision polish of the winning hit (RT_FLAG_NO_REFINE switches it off)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
c = api.Context(0)
c.upload_scene(S.flatten(S.rtiow_scene()))
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
c.set_camera(**S.rtiow_camera(float(np.float32(16 / 9))).desc())
for flags in (0, api.RT_FLAG_NO_REFINE):
    ms = []
    for _ in range(3):
        c.render(api.Context.params(1920, 1080, 100, 50, 1, flags=flags), buf.data_ptr()); ms.append(c.stats()["kernel_ms"])
    print("flags", flags, "C2", round(min(ms), 2), "ms")
for name in ("image_120x80_16spp.npz", "image_240x160_64spp.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    W, H, spp = int(g["W"]), int(g["H"]), int(g["spp"])
    c.set_camera(**S.rtiow_camera(W / H).desc())
    for flags in (0, api.RT_FLAG_NO_REFINE):
        img = c.render_host(api.Context.params(W, H, spp, int(g["depth"]), int(g["seed"]), flags=flags)); st = c.stats()
        mean = img[..., :3].astype(np.float64) / spp
        print(name, "flags", flags, "rmse %.3e" % np.sqrt(np.mean((mean - g["sums"] / spp) ** 2)), "ray diff", st["rays"] - int(g["rays"]),
              "pixels off > 1e-5: %.2f%%" % (100 * (np.abs(mean - g["sums"] / spp).max(axis=2) > 1e-5).mean()))
g = np.load(os.path.join(ROOT, "tests", "golden", "primary_rays.npz"))
c.set_camera(**S.rtiow_camera(1.5).desc())
