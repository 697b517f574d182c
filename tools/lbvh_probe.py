"""Build time and render time of the two BVH builders on the 10^6-sphere scene."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
n = int(os.environ.get("C4_SPHERES", "1000000"))
aspect = float(np.float32(16 / 9))
flat = S.stress_flat(n)
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
out = {}
ref = None
for name, accel in (("host_sah", api.RT_ACCEL_BVH), ("device_lbvh", api.RT_ACCEL_BVH_DEVICE)):
    c = api.Context(0)
    for rep in range(2):
        t0 = time.perf_counter(); c.upload_scene(flat, accel=accel); up = time.perf_counter() - t0
    c.set_camera(**S.stress_camera(aspect).desc())
    c.render(api.Context.params(3840, 2160, 16, 50, 1), buf.data_ptr()); st = c.stats()
    img = buf.cpu().numpy().copy()
    if ref is None: ref = img
    c.render(api.Context.params(3840, 2160, 4, 50, 1, flags=api.RT_FLAG_COUNTERS), buf.data_ptr()); sc = c.stats()
    out[name] = dict(upload_s=up, build_ms=st["build_ms"], nodes=st["bvh_nodes"], depth=st["bvh_depth"], render_ms_16spp=st["kernel_ms"],
                     mrays_s=st["rays"] / st["kernel_ms"] / 1e3, box_per_ray=sc["box_tests"] / sc["rays"], sph_per_ray=sc["sphere_tests"] / sc["rays"],
                     same_image=bool(np.array_equal(img, ref)))
    print(name, json.dumps(out[name]), flush=True)
    c.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "lbvh_probe.json"), "w"), indent=1)
