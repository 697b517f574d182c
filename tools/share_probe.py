"""Frame time of the headline configuration by the subtree-sharing threshold (rt_render_params.share_below)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytrace_amd import api, scene as S
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_BVH)
ctx.set_camera(**S.rtiow_camera(float(np.float32(16 / 9))).desc())
buf = torch.zeros((1080, 1920, 4), device="cuda")
ref = None
for share in [int(a) for a in sys.argv[1:]] or [-1, 4, 8, 12, 16, 24, 32, 48, 64]:
    best = None
    for _ in range(4):
        ctx.render(api.Context.params(1920, 1080, 100, 50, 1, share_below=share), buf.data_ptr())
        st = ctx.stats()
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    img = buf.cpu().numpy()
    if ref is None:
        ref = img
    ctx.render(api.Context.params(1920, 1080, 100, 50, 1, share_below=share, flags=api.RT_FLAG_COUNTERS), buf.data_ptr())
    c = ctx.stats()
    print(f"share_below {share:3d}: {best:7.3f} ms  same bits {np.array_equal(img, ref)}  rays {st['rays']}  box tests/ray {c['box_tests'] / c['rays']:.2f}"
          f"  sphere tests/ray {c['sphere_tests'] / c['rays']:.2f}  traversal lane util {c['phase_clocks'][0] / (64.0 * c['phase_clocks'][1]):.3f}", flush=True)
