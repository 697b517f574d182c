"""Developer probe: per-ray work counters and timing of render variants on the GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytrace_amd import api, scene as S

W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "100"))
flat = S.flatten(S.rtiow_scene())
c = api.Context(0)
c.upload_scene(flat, accel=api.RT_ACCEL_BVH)
c.set_camera((6, 3, -13), (0, 0, 0), (0, 1, 0), 20.0, float(np.float32(W / H)), 0.1, 10.0)
P = api.Context.params
import torch
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
def run(label, reps=3, **kw):
    ms = []
    for _ in range(reps):
        c.render(P(W, H, SPP, 50, 1, **kw), buf.data_ptr())
        st = c.stats(); ms.append(st["kernel_ms"])
    print(f"{label:28s} min {min(ms):8.2f} ms  {st['rays'] / min(ms) / 1e3:9.1f} Mrays/s  K={st['lanes_per_pixel']}")
    return st
st = run("default")
run("no LDS scene", flags=api.RT_FLAG_NO_LDS_SCENE)
a = c.render_host(P(256, 128, 8, 50, 3)); d2 = c.render_host(P(256, 128, 8, 50, 3, flags=api.RT_FLAG_NO_LDS_SCENE)); print("LDS == HBM scene bit-exact:", np.array_equal(a, d2))
run("no refine", flags=api.RT_FLAG_NO_REFINE)
for k in (2, 4, 8):
    run(f"K={k}", lanes_per_pixel=k)
cst = run("counters", reps=1, flags=api.RT_FLAG_COUNTERS)
r = cst["rays"]
print("traversal: lane steps/ray %.2f, wave steps %.3g, lane utilisation %.3f" % (cst["phase_clocks"][0] / r, cst["phase_clocks"][1], cst["phase_clocks"][0] / (64.0 * cst["phase_clocks"][1])))
print({k: round(v / r, 3) for k, v in cst.items() if k.endswith("tests") or k in ("sphere_hits", "shades", "samples")})
pst = run("phase clocks", reps=1, flags=api.RT_FLAG_PHASE_CLOCKS)
pc = np.array(pst["phase_clocks"], dtype=np.float64)
print("phase share: camera %.3f always %.3f traversal %.3f record %.3f scatter %.3f" % tuple(pc / pc.sum()))
