"""Prints the measured parity figures against the golden images (for the documentation)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytrace_amd import api, scene as S
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()))
for name in ("image_120x80_16spp.npz", "image_240x160_64spp.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    W, H, spp = int(g["W"]), int(g["H"]), int(g["spp"])
    ctx.set_camera(**S.rtiow_camera(W / H).desc())
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=accel)
        img = ctx.render_host(api.Context.params(W, H, spp, int(g["depth"]), int(g["seed"])))
        st = ctx.stats()
        mean = img[..., :3].astype(np.float64) / spp
        rmse = np.sqrt(np.mean((mean - g["sums"] / spp) ** 2))
        differing = (np.abs(mean - g["sums"] / spp).max(axis=2) > 1e-5).mean()
        print(f"{name} accel {accel}: rmse {rmse:.3e} rays gpu {st['rays']} ref {int(g['rays'])} diff {st['rays'] - int(g['rays'])} pixels differing >1e-5: {100*differing:.2f}%")
