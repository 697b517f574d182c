"""Shares of a wave's time by phase of the path loop (the phase-clock build of the all-features kernel: unit
hand-out and camera rays / always-tested primitives and the scan / tree walk / shading), for the four bench scenes
at reduced sample counts.  The production kernels are specialised builds; the shares are a guide, not a measurement of them.
usage: python tools/phase_shares.py [--lib other/librt_amd.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

if "--lib" in sys.argv:                           # another build of the library
    api.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
A = float(np.float32(16 / 9))
room = dict(background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
cases = [("default world @100", lambda: S.flatten(S.rtiow_scene()), S.rtiow_camera(A), api.RT_ACCEL_BVH, 100),
         ("Cornell @100", lambda: S.flatten(S.cornell_scene(), **room), S.cornell_camera(A), api.RT_ACCEL_LIST, 100),
         ("room of turned boxes @100", lambda: S.flatten(S.cornell_boxes_scene(), **room), S.cornell_camera(A), api.RT_ACCEL_LIST, 100)]
for name, make, cam, accel, spp in cases:
    ctx = api.Context(0)
    ctx.upload_scene(make(), accel=accel)
    ctx.set_camera(**cam.desc())
    buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
    p = api.Context.params(1920, 1080, spp, 50, 1)
    pp = api.Context.params(1920, 1080, spp, 50, 1, flags=api.RT_FLAG_PHASE_CLOCKS)
    for _ in range(2):
        ctx.render(p, buf.data_ptr()); ms = ctx.stats()["kernel_ms"]
    ctx.render(pp, buf.data_ptr())
    st = ctx.stats()
    ph = st["phase_clocks"]
    tot = float(sum(ph)) or 1.0
    print(f"{name:28s} {ms:8.2f} ms (phase build {st['kernel_ms']:8.2f})  rays {st['rays'] / 1e6:8.1f} M  hand-out+camera {ph[0] / tot:5.1%}  "
          f"draws (kernels that draw in one place) {ph[3] / tot:5.1%}  always/scan {ph[1] / tot:5.1%}  walk {ph[2] / tot:5.1%}  shading {ph[4] / tot:5.1%}", flush=True)
    ctx.close()
