"""Ray-count difference (kernels minus checker) over many fuzz scenes: is it centred on zero?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _random_scene, _random_field
from raytrace_amd import api, scene as S
from oracle.pyoracle import Oracle
oracle = Oracle("f64")
lo, hi = int(os.environ.get("LO", "0")), int(os.environ.get("HI", "1000"))
W, H, spp = 72, 48, 8
diffs, rels = [], []
ctx = api.Context(0)
for seed in range(lo, hi):
    if os.environ.get("FIELD"):
        rng = np.random.default_rng(50000 + seed)
        flat, cam, scale, offset, side = _random_field(S, rng)
    else:
        rng = np.random.default_rng(1000 + seed)
        flat, cam, scale, offset = _random_scene(S, rng)
    cd = oracle.camera(**cam)
    _, n_o = oracle.render(flat, cd, W, H, spp, 50, 77 + seed, threads=8)
    ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH); ctx.set_camera(**cam)
    ctx.render_host(api.Context.params(W, H, spp, 50, 77 + seed))
    n_g = ctx.stats()["rays"]
    diffs.append(n_g - n_o); rels.append((n_g - n_o) / n_o)
    if abs(n_g - n_o) > 40: print("seed", seed, "gpu", n_g, "ref", n_o, "diff", n_g - n_o, flush=True)
d = np.asarray(diffs); r = np.asarray(rels)
print("seeds", len(d), "mean diff", d.mean(), "median", np.median(d), "positive", int((d > 0).sum()), "negative", int((d < 0).sum()), "zero", int((d == 0).sum()))
print("mean relative", r.mean(), "sd of relative", r.std(), "max |rel|", np.abs(r).max())
