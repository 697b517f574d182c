import os, sys, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
c = api.Context(0)
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
def run(label, W, H, spp, K=0):
    ms = []
    for _ in range(3):
        c.render(api.Context.params(W, H, spp, 50, 1, lanes_per_pixel=K), buf.data_ptr()); st = c.stats(); ms.append(st["kernel_ms"])
    print("%%s %%s: %%.2f ms %%.0f Mrays/s" %% (os.environ["VARIANT"], label, min(ms), st["rays"] / min(ms) / 1e3), flush=True)
if os.environ.get("RT_PROBE") == "c4":
    c.upload_scene(S.stress_flat(1000000), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
    run("C4@16", 3840, 2160, 16)
    run("C4@64", 3840, 2160, 64)
    c.upload_scene(S.stress_flat(100000), accel=api.RT_ACCEL_BVH)
    run("10^5 spheres 1080p@64", 1920, 1080, 64)
    sys.exit(0)
c.upload_scene(S.flatten(S.rtiow_scene())); c.set_camera(**S.rtiow_camera(aspect).desc())
for K in (4, 8):
    run("C2 K%%d" %% K, 1920, 1080, 100, K)
c.upload_scene(S.flatten(S.rtiow_scene()), accel=api.RT_ACCEL_LIST)
for K in (4, 8):
    run("C2 list@20 K%%d" %% K, 1920, 1080, 20, K)
c.upload_scene(S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID)); c.set_camera(**S.cornell_camera(aspect).desc())
for K in (4, 8):
    run("C3 K%%d" %% K, 1920, 1080, 100, K)
c.upload_scene(S.stress_flat(1000000), accel=api.RT_ACCEL_BVH); c.set_camera(**S.stress_camera(aspect).desc())
for K in (4, 8):
    run("C4@16 K%%d" %% K, 3840, 2160, 16, K)
''' % ROOT
lib = os.path.join(ROOT, "raytrace_amd", "librt_amd.so")
keep = lib + ".keep"
shutil.copy(lib, keep)
try:
    for v in sorted(os.listdir(os.path.join(ROOT, "build_variants"))):
        shutil.copy(os.path.join(ROOT, "build_variants", v), lib)
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VARIANT=v), check=False, timeout=300)
finally:
    shutil.copy(keep, lib); os.remove(keep)
