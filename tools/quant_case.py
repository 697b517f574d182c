"""One configuration of tools/quant_probe.py, for a profiler: the mixed-scale scene rendered three
times through the quantised (default) or the float nodes (--float).  usage: python tools/quant_case.py [--float] [--spp N] [--depth D]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytrace_amd import api, scene as S          # noqa: E402
from test_gpu_parity_paths import mixed_scale_scene   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--float", action="store_true")
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--depth", type=int, default=1)
a = ap.parse_args()
flat, cam, _ = mixed_scale_scene(S, 0.0, 0.15)
ctx = api.Context(0)
ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH)
ctx.set_camera(**cam)
flags = api.RT_FLAG_NO_LDS_SCENE | (api.RT_FLAG_FLOAT_NODES if a.float else 0)
for it in range(3):
    ctx.render_host(api.Context.params(1536, 1024, a.spp, a.depth, 5, flags=flags))
    print(ctx.stats()["kernel_ms"], flush=True)
ctx.close()
