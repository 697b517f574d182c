"""Times the walk of a tree in HBM through its quantised and its float nodes on the mixed-scale
scene of tests/test_gpu_parity_paths.py and on the stress scene at several frame sizes.  usage: python tools/quant_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytrace_amd import api, scene as S          # noqa: E402
from test_gpu_parity_paths import mixed_scale_scene   # noqa: E402


def run(name, flat, cam, W, H, spp, depth):
    for accel, aname in ((api.RT_ACCEL_BVH, "host"), (api.RT_ACCEL_BVH_DEVICE, "device")):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        out = {}
        for label, flags in (("quantised", api.RT_FLAG_NO_LDS_SCENE), ("float", api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES)):
            ms = []
            for it in range(3):
                img = ctx.render_host(api.Context.params(W, H, spp, depth, 5, flags=flags))
                ms.append(ctx.stats()["kernel_ms"])
            ctx.render_host(api.Context.params(W, H, spp, depth, 5, flags=flags | api.RT_FLAG_COUNTERS))
            st = ctx.stats()
            out[label] = (min(ms), st["box_tests"] / st["rays"], img)
        st = ctx.stats()
        same = np.array_equal(out["quantised"][2], out["float"][2])
        print(f"{name} {aname}-built: nodes {st['bvh_nodes']} coarse {st['bvh_coarse_nodes']}  quantised {out['quantised'][0]:.2f} ms "
              f"({out['quantised'][1]:.1f} box tests/ray)  float {out['float'][0]:.2f} ms ({out['float'][1]:.1f})  same image {same}", flush=True)
        ctx.close()


if __name__ == "__main__":
    flat, cam, _ = mixed_scale_scene(S, 0.0, 0.15)
    run("mixed scales", flat, cam, 1536, 1024, 16, 1)
    sf = S.stress_flat(1_000_000)
    for W, H, spp in ((1920, 1080, 8), (1920, 1080, 64), (3840, 2160, 16), (3840, 2160, 64)):
        run(f"stress {W}x{H}@{spp}", sf, S.stress_camera(16 / 9).desc(), W, H, spp, 50)
