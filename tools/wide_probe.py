"""Binary against four-wide tree on the headline frame (tree in LDS) and on the 10^6-sphere frame (tree in
HBM): frame time, node tests per ray, same bits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytrace_amd import api, scene as S

def run(name, flat, cam, W, H, spp, accels, reps=3, flags=0):
    buf = torch.zeros((H, W, 4), device="cuda")
    ref = None
    for label, accel in accels:
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        best = None
        for _ in range(reps):
            ctx.render(api.Context.params(W, H, spp, 50, 1, flags=flags), buf.data_ptr())
            st = ctx.stats()
            best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
        img = buf.cpu().numpy()
        if ref is None:
            ref = img
        ctx.render(api.Context.params(W, H, min(spp, 16), 50, 1, flags=flags | api.RT_FLAG_COUNTERS), buf.data_ptr())
        c = ctx.stats()
        print(f"{name} {label:12s}: {best:8.3f} ms  same bits {np.array_equal(img, ref)}  accel {st['accel']} nodes {st['bvh_nodes']} depth {st['bvh_depth']}"
              f"  build {st['build_ms']:.0f} ms  box tests/ray {c['box_tests'] / c['rays']:.2f}  node steps/ray {(c['phase_clocks'][0]) / c['rays']:.2f}"
              f"  sphere tests/ray {c['sphere_tests'] / c['rays']:.2f}  lane util {c['phase_clocks'][0] / (64.0 * c['phase_clocks'][1]):.3f}", flush=True)
        ctx.close()

aspect = float(np.float32(16 / 9))
which = sys.argv[1:] or ["c2", "c4"]
if "c2" in which:
    run("c2", S.flatten(S.rtiow_scene()), S.rtiow_camera(aspect).desc(), 1920, 1080, 100,
        [("bvh2", api.RT_ACCEL_BVH), ("bvh4", api.RT_ACCEL_BVH4), ("bvh2 dev", api.RT_ACCEL_BVH_DEVICE), ("bvh4 dev", api.RT_ACCEL_BVH4_DEVICE)])
    run("c2 hbm", S.flatten(S.rtiow_scene()), S.rtiow_camera(aspect).desc(), 1920, 1080, 100,
        [("bvh2", api.RT_ACCEL_BVH), ("bvh4", api.RT_ACCEL_BVH4)], flags=api.RT_FLAG_NO_LDS_SCENE)
if "c4" in which:
    run("c4", S.stress_flat(1_000_000), S.stress_camera(aspect).desc(), 3840, 2160, 16,
        [("bvh2", api.RT_ACCEL_BVH), ("bvh4", api.RT_ACCEL_BVH4), ("bvh2 dev", api.RT_ACCEL_BVH_DEVICE), ("bvh4 dev", api.RT_ACCEL_BVH4_DEVICE)], reps=2)
