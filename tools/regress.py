"""Frame times of every kernel family on fixed scenes, one launch at a time: the table a change to the
kernels is held against besides the four bench configurations (a change made for one scene has cost
another its speed before), with a digest of every frame: two builds whose digests agree render the same
bits.  usage: python tools/regress.py [--lib other/librt_amd.so] [--quick] [out.json]"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

ARGS = sys.argv[1:]
if "--lib" in ARGS:                               # another build of the library (same-box A/B)
    api.LIB_PATH = os.path.abspath(ARGS[ARGS.index("--lib") + 1])
    del ARGS[ARGS.index("--lib"):ARGS.index("--lib") + 2]
QUICK = "--quick" in ARGS                         # without the rows that take seconds each
if QUICK:
    ARGS.remove("--quick")

A = 16 / 9
RTIOW_CAM = S.rtiow_camera(A).desc()
rows = []


def measure(name, flat, cam, W, H, spp, depth=50, accel=api.RT_ACCEL_AUTO, flags=0, reps=3, ctx=None, **kw):
    own = ctx is None
    if own:
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
    ctx.set_camera(**cam)
    # (integer sums take 32 bytes a pixel, the float image 16)
    buf = torch.zeros((H, W, 4), dtype=torch.int64 if flags & api.RT_FLAG_FIXED_SUMS else torch.float32, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        ctx.render(api.Context.params(W, H, spp, depth, 1, flags=flags, **kw), buf.data_ptr())
        st = ctx.stats()
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    digest = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]
    rows.append({"case": name, "frame": f"{W}x{H}@{spp}", "kernel_ms": round(best["kernel_ms"], 3),
                 "mrays_s": round(best["rays"] / best["kernel_ms"] / 1e3, 1), "bvh_depth": best["bvh_depth"],
                 "rays": best["rays"], "digest": digest})
    print(f"{name:58s} {W}x{H}@{spp:<5d} {best['kernel_ms']:9.3f} ms {best['rays'] / best['kernel_ms'] / 1e3:9.1f} Mrays/s  {digest}", flush=True)
    if own:
        ctx.close()
    return ctx


rtiow = S.flatten(S.rtiow_scene())
c = api.Context(0)
c.upload_scene(rtiow, accel=api.RT_ACCEL_BVH)
measure("rtiow, tree in LDS", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c)
measure("rtiow, tree in LDS, 4 lanes per pixel", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c, lanes_per_pixel=4)
measure("rtiow, tree out of HBM (32-byte nodes)", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c, flags=api.RT_FLAG_NO_LDS_SCENE)
measure("rtiow, tree out of HBM (64-byte float nodes)", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c,
        flags=api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES)
measure("rtiow, tree in LDS, integer sums out", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c, flags=api.RT_FLAG_FIXED_SUMS)
measure("rtiow, one eighth of the rows (interleaved 4-row tiles)", rtiow, RTIOW_CAM, 1920, 1080, 100, ctx=c, tiles=(4, 3, 8))
c.close()
measure("rtiow, linear scan (scalar loads)", rtiow, RTIOW_CAM, 1920, 1080, 100, accel=api.RT_ACCEL_LIST)
measure("rtiow, GPU-built tree", rtiow, RTIOW_CAM, 1920, 1080, 100, accel=api.RT_ACCEL_BVH_DEVICE)

cornell = S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID)
measure("cornell (scan of 2 spheres from LDS + 6 rects)", cornell, S.cornell_camera(A).desc(), 1920, 1080, 100)
if not QUICK:
    measure("cornell at the sample count of configs[2]", cornell, S.cornell_camera(A).desc(), 1920, 1080, 1000)
boxes = S.flatten(S.cornell_boxes_scene(), background=S.RT_BG_SOLID)
measure("cornell with two turned boxes (instance rects)", boxes, S.cornell_camera(A).desc(), 1920, 1080, 100)
sky = S.flatten(S.skybox_scene(), background=S.RT_BG_SOLID)
measure("skybox world (image textures, boxes, 125 balls)", sky, S.skybox_camera(A).desc(), 1920, 1080, 100)

for n, (W, H, spp) in ((20_000, (1920, 1080, 64)), (100_000, (1920, 1080, 64)), (1_000_000, (3840, 2160, 16))):
    flat = S.stress_flat(n)
    cam = S.stress_camera(A).desc()
    c = api.Context(0)
    c.upload_scene(flat, accel=api.RT_ACCEL_BVH)
    measure(f"{n} spheres, host tree, 32-byte nodes", flat, cam, W, H, spp, ctx=c)
    if n == 1_000_000 and not QUICK:
        measure(f"{n} spheres at the sample count of configs[3]", flat, cam, W, H, 64, ctx=c)
    measure(f"{n} spheres, host tree, 64-byte float nodes", flat, cam, W, H, spp, ctx=c, flags=api.RT_FLAG_FLOAT_NODES)
    c.close()
    measure(f"{n} spheres, GPU-built tree", flat, cam, W, H, spp, accel=api.RT_ACCEL_BVH_DEVICE)

if ARGS:
    with open(ARGS[0], "w") as f:
        json.dump(rows, f, indent=1)
