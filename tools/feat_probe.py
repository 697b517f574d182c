"""Frame time of the kernel built for what a scene holds against the kernel that carries the code of every
feature (RT_FLAG_ALL_FEATURES), same box, same scene: Cornell @1000, the 10^6-sphere scene @64, a textured
room, the default world.  usage: python tools/feat_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

A = float(np.float32(16 / 9))
CASES = [
    ("default world (spheres)", S.flatten(S.rtiow_scene()), S.rtiow_camera(A).desc(), 1920, 1080, 100, api.RT_ACCEL_BVH),
    ("cornell (rects)", S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID), S.cornell_camera(A).desc(), 1920, 1080, 1000, api.RT_ACCEL_AUTO),
    ("skybox world (rects + textures + instances)", S.flatten(S.skybox_scene(), background=S.RT_BG_SOLID), S.skybox_camera(A).desc(), 1920, 1080, 100, api.RT_ACCEL_AUTO),
    ("10^6 spheres (textures)", S.stress_flat(1_000_000), S.stress_camera(A).desc(), 3840, 2160, 64, api.RT_ACCEL_BVH),
]
for name, flat, cam, W, H, spp, accel in CASES:
    ctx = api.Context(0)
    ctx.upload_scene(flat, accel=accel)
    ctx.set_camera(**cam)
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    out = {"own kernel": 1e9, "all features": 1e9}
    for _ in range(4):                            # interleaved: the two do not see different clocks or temperatures
        for label, flags in (("own kernel", 0), ("all features", api.RT_FLAG_ALL_FEATURES)):
            ctx.render(api.Context.params(W, H, spp, 50, 1, flags=flags), buf.data_ptr())
            out[label] = min(out[label], ctx.stats()["kernel_ms"])
    print(f"{name:46s} {W}x{H}@{spp:<5d} own kernel {out['own kernel']:9.3f} ms   all features {out['all features']:9.3f} ms", flush=True)
    ctx.close()
