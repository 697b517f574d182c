"""Four-wide quantised nodes against the binary quantised ones on the stress scene (tree walked out of
HBM): same frame bits, frame time, counters.  usage: python tools/wide_probe.py [spp] [n_spheres]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W, H = 3840, 2160
flat = S.stress_flat(n)
ref = None
for cap in (0, 23, 28):
    ctx = api.Context(0)
    api.lib().rt_tune_wide_stack(ctx._h, ctypes.c_int(cap))
    ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH)
    ctx.set_camera(**S.stress_camera(16 / 9).desc())
    buf = torch.zeros((H, W, 4), device="cuda")
    st = ctx.stats()
    print(f"cap {cap}: binary nodes {st['bvh_nodes']} depth {st['bvh_depth']}; wide nodes {st['bvh_wide_nodes']} "
          f"depth {st['bvh_wide_depth']} stack {st['bvh_wide_stack']}; build {st['build_ms']:.0f} ms", flush=True)
    for name, fl in (("binary", 0), ("wide", api.RT_FLAG_WIDE_NODES)):
        if name == "binary" and ref is not None:
            continue
        for sl, q in ((0, 0), (32, 4), (24, 4), (48, 2), (32, 2), (64, 4)):
            best = None
            for _ in range(2):
                ctx.render(api.Context.params(W, H, spp, 50, 1, flags=fl, traversal_slice=sl, leaf_quorum=q), buf.data_ptr())
                ms = ctx.stats()["kernel_ms"]
                best = ms if best is None else min(best, ms)
            print(f"  {name:6s} slice {sl:3d} quorum {q}: {best:8.3f} ms", flush=True)
        torch.cuda.synchronize()
        img = buf.clone()
        if ref is None:
            ref = img
        else:
            print(f"  {name}: frame equals the binary walk's: {bool(torch.equal(img, ref))}", flush=True)
        ctx.render(api.Context.params(W, H, spp, 50, 1, flags=fl | api.RT_FLAG_COUNTERS), buf.data_ptr())
        torch.cuda.synchronize()
        st = ctx.stats()
        print(f"  {name}: box tests/ray {st['box_tests'] / st['rays']:.1f} sphere tests/ray {st['sphere_tests'] / st['rays']:.2f} "
              f"rays {st['rays']}", flush=True)
    del ctx
