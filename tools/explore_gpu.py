"""Developer probe: prints parity metrics and timings of the render path on the GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from raytrace_amd import api, scene as S

flat = S.flatten(S.rtiow_scene())
CAM = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.5, aperture=0.1, focus_dist=10.0)
g = load_golden("primary_rays.npz")
for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
    c = api.Context(0)
    t0 = time.time(); c.upload_scene(flat, accel=accel); print("accel", accel, "upload s", time.time() - t0)
    c.set_camera(**CAM)
    h = c.trace_primary(g["rays"], 0.001)
    bad = np.where(h["id"] != g["id"])[0]
    print(" id mismatches", len(bad), bad[:10], g["kind"][bad][:10], h["id"][bad][:10], g["id"][bad][:10])
    hit = (g["id"] >= 0) & (h["id"] == g["id"])
    trel = np.abs(h["t"][hit] - g["t"][hit]) / np.abs(g["t"][hit])
    print(" t rel max", trel.max(), "n abs max", np.abs(h["n"][hit] - g["n"][hit]).max(),
          "front mismatches", int((h["front_face"][hit] != g["front"][hit]).sum()))
    for name in ("image_120x80_16spp.npz", "image_240x160_64spp.npz"):
        gi = load_golden(name)
        W, H, spp = int(gi["W"]), int(gi["H"]), int(gi["spp"])
        img = c.render_host(api.Context.params(W, H, spp, 50, int(gi["seed"])))
        st = c.stats()
        mean = img[..., :3].astype(np.float64) / spp
        print(" ", name, "rmse", np.sqrt(np.mean((mean - gi["sums"] / spp) ** 2)), "rays", st["rays"], int(gi["rays"]),
              "ms", st["kernel_ms"], "K", st["lanes_per_pixel"])
    for (W, H, spp) in ((1920, 1080, 10), (1920, 1080, 100)):
        c.set_camera(**dict(CAM, aspect=W / H))
        buf = c.render_host(api.Context.params(W, H, spp, 50, 1))
        st = c.stats()
        print(f"  {W}x{H}@{spp}: {st['kernel_ms']:.2f} ms, rays {st['rays']}, {st['rays'] / st['kernel_ms'] / 1e3:.1f} Mrays/s, bvh {st['bvh_nodes']} nodes depth {st['bvh_depth']}")
    c.close()
