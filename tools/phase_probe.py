"""Shares of a wave's time by phase (RT_FLAG_PHASE_CLOCKS build of the frame kernel) for a bench
configuration: hand-out and camera rays, always-tested primitives, closest hit (tree walk or scan),
record, shading.  usage: python tools/phase_probe.py <c2|c3|c4> [spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                      # noqa: E402
from raytrace_amd import api                      # noqa: E402
from bench import Workload                        # noqa: E402

wl = Workload(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
ctx = api.Context(0)
ctx.upload_scene(wl.flat, accel=wl.accel)
ctx.set_camera(**wl.camera.desc())
buf = torch.zeros((wl.H, wl.W, 4), device="cuda")
for flags in (0, api.RT_FLAG_PHASE_CLOCKS):
    for _ in range(2):
        ctx.render(api.Context.params(wl.W, wl.H, wl.spp, wl.depth, 1, flags=flags), buf.data_ptr())
        st = ctx.stats()
    print("flags", flags, "kernel_ms", st["kernel_ms"], "rays", st["rays"])
ph = st["phase_clocks"]
tot = float(sum(ph))
print("phase shares: hand-out+camera %.3f  always-tested %.3f  closest hit %.3f  (unused %.3f)  shading %.3f" % tuple(p / tot for p in ph))
