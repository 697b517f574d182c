"""Renders the headline frame a few times, one launch at a time, with the library named by --lib (default:
the one in the tree): the program PMC passes and kernel traces of a build variant are taken on.
usage: rocprofv3 --pmc ... -- python3 tools/frame_loop.py [--lib build_variants/librt_X.so] [--frames N] [--no-lds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

A = sys.argv[1:]
if "--lib" in A:
    api.LIB_PATH = os.path.abspath(A[A.index("--lib") + 1])
frames = int(A[A.index("--frames") + 1]) if "--frames" in A else 5
flags = api.RT_FLAG_NO_LDS_SCENE if "--no-lds" in A else 0
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.rtiow_camera(float(np.float32(16 / 9))).desc())
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
ms = []
for _ in range(frames):
    ctx.render(api.Context.params(1920, 1080, 100, 50, 1, flags=flags), buf.data_ptr())
    ms.append(ctx.stats()["kernel_ms"])
print("kernel_ms", " ".join(f"{m:.3f}" for m in ms), "rays", ctx.stats()["rays"])
ctx.close()
