"""One fuzz scene (tests/test_gpu_fuzz.py) rendered by the checker and by the kernels: where do
the two frames and ray counts differ?  usage: SEED=2161 python tools/fuzz_rays.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _random_scene, _random_field
from raytrace_amd import api, scene as S
from oracle.pyoracle import Oracle
oracle = Oracle("f64")
seed = int(os.environ.get("SEED", "2161"))
if os.environ.get("FIELD"):
    rng = np.random.default_rng(50000 + seed)
    flat, cam, scale, offset, side = _random_field(S, rng)
else:
    rng = np.random.default_rng(1000 + seed)
    flat, cam, scale, offset = _random_scene(S, rng)
cd = oracle.camera(**cam)
W, H, spp = 72, 48, 8
ctx = api.Context(0); ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH); ctx.set_camera(**cam)
print("scale", scale, "offset", offset, "prims", len(flat.prim_type), "types", np.bincount(np.asarray(flat.prim_type)))
tot_g = tot_o = 0
for s in range(spp):
    sums, n_o = oracle.render(flat, cd, W, H, 1, 50, 77 + seed, threads=8, sample_begin=s) if "sample_begin" in oracle.render.__code__.co_varnames else (None, None)
    if sums is None:
        break
    img = ctx.render_host(api.Context.params(W, H, 1, 50, 77 + seed, sample_begin=s))
    n_g = ctx.stats()["rays"]
    d = np.abs(img[..., :3] - sums).max(axis=2)
    print(f"sample {s}: rays gpu {n_g} ref {n_o} diff {n_g - n_o}; pixels off by > 1e-5: {(d > 1e-5).sum()} max {d.max():.3g}")
    tot_g += n_g; tot_o += n_o
print("total", tot_g, tot_o)
# which pixels, which primitive is seen there, and from which depth limit on the frames differ
s = int(os.environ.get("SAMPLE", "2"))
full_o, _ = oracle.render(flat, cd, W, H, 1, 50, 77 + seed, threads=8, sample_begin=s)
full_g = ctx.render_host(api.Context.params(W, H, 1, 50, 77 + seed, sample_begin=s))[..., :3]
bad = np.argwhere(np.abs(full_g - full_o).max(axis=2) > 1e-5)
per_depth = {}
for D in list(range(1, 12)) + [15, 20, 30, 40, 49]:
    o_, no = oracle.render(flat, cd, W, H, 1, D, 77 + seed, threads=8, sample_begin=s)
    g_ = ctx.render_host(api.Context.params(W, H, 1, D, 77 + seed, sample_begin=s))[..., :3]
    per_depth[D] = (np.abs(g_ - o_).max(axis=2) > 1e-5, ctx.stats()["rays"] - no)
for (j, i) in bad:
    first = next((D for D in sorted(per_depth) if per_depth[D][0][j, i]), None)
    ray = oracle.get_ray(cd, (i + 0.5) / (W - 1), (j + 0.5) / (H - 1), 5, 0, 0)[0]
    hit = oracle.trace(flat, np.asarray([ray]), t_min=0.001) if ray is not None else None
    pid = int(hit["id"][0]) if hit is not None else -2
    print(f"pixel ({i},{j}) differs from depth limit {first}; centre ray sees prim {pid}"
          + (f" type {flat.prim_type[pid]} mat {flat.prim_mat[pid] if hasattr(flat, 'prim_mat') else '?'} geom {np.round(flat.prim_geom[pid][:6], 3)}" if pid >= 0 else ""))
import collections
hist = collections.Counter(next((D for D in sorted(per_depth) if per_depth[D][0][j, i]), None) for (j, i) in bad)
print("first differing depth limit -> pixels:", sorted(hist.items(), key=lambda kv: (kv[0] is None, kv[0])))
print("ray count diff by depth limit:", {D: int(v[1]) for D, v in per_depth.items()})
print("aperture", cam["aperture"])
