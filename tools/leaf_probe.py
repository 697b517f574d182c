"""C2 frame time against the host builder's leaf size and SAH node cost."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
flat = S.flatten(S.rtiow_scene())
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
ref = None
for leaf_max, cost in ((4, 0), (2, 0), (3, 0), (6, 0), (8, 0), (4, 1.0), (4, 2.0), (8, 1.0), (8, 1.5), (8, 2.0), (8, 3.0), (8, 4.0), (6, 2.0)):
    os.environ["RT_TUNE_LEAF_MAX"] = str(leaf_max); os.environ["RT_TUNE_NODE_COST"] = str(cost)
    c = api.Context(0)
    c.upload_scene(flat, accel=api.RT_ACCEL_BVH); c.set_camera(**S.rtiow_camera(aspect).desc())
    ms = []
    for _ in range(3):
        c.render(api.Context.params(1920, 1080, 100, 50, 1), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
    img = buf.cpu().numpy().copy()
    if ref is None: ref = img
    c.render(api.Context.params(1920, 1080, 8, 50, 1, flags=api.RT_FLAG_COUNTERS), buf.data_ptr()); sc = c.stats()
    r = sc["rays"]
    print(f"leaf<= {leaf_max} node_cost {cost}: {min(ms):6.2f} ms nodes {st['bvh_nodes']} depth {st['bvh_depth']} box/ray {sc['box_tests']/r:.2f} sph/ray {sc['sphere_tests']/r:.2f} steps {sc['phase_clocks'][0]/r:.2f} util {sc['phase_clocks'][0]/max(1,sc['phase_clocks'][1])/64:.3f} same {np.array_equal(img, ref)}", flush=True)
    c.close()
