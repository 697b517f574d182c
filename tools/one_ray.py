"""One ray of one fuzz scene under the kernels (polished and unpolished) and the checker."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _random_scene
from raytrace_amd import api, scene as S
from oracle.pyoracle import Oracle
oracle = Oracle("f64")
seed = int(os.environ.get("SEED", "14114"))
rng = np.random.default_rng(1000 + seed)
flat, cam, scale, offset = _random_scene(S, rng)
ray = np.array([[float(x) for x in os.environ.get("RAY", "1967.32373047 1595.25219727 5957.82275391 469.53118896 -890.46380615 -5529.64501953").split()]], dtype=np.float32).astype(np.float64)
w = oracle.trace(flat, ray, t_min=0.001)
print("checker id", w["id"][0], "t %.12g" % w["t"][0], "n", w["n"][0])
for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
    ctx = api.Context(0); ctx.upload_scene(flat, accel=accel)
    for kw in ({}, {"refine": False}):
        try:
            h = ctx.trace_primary(ray, **kw)
        except TypeError:
            continue
        print("accel", accel, kw, "id", h["id"][0], "t %.12g" % h["t"][0], "n", h["n"][0], "p", h["p"][0])
    ctx.close()
