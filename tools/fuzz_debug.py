import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _random_scene, _unit
from raytrace_amd import api, scene as S
from oracle.pyoracle import Oracle
oracle = Oracle("f64")
for seed in (int(os.environ.get("SEED", "281")),):
    rng = np.random.default_rng(1000 + seed)
    flat, cam, scale, offset = _random_scene(S, rng)
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 5, k, 0)[0] for k in range(1500)]
    for _ in range(1500):
        o = offset + rng.uniform(-6, 6, 3) * scale * [1, 0.3, 1] + [0, 1.5 * scale, 0]
        rays.append(np.concatenate([o, _unit(rng.normal(size=3)) * rng.uniform(0.3, 2)]))
    rays = np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)
    want = oracle.trace(flat, rays, t_min=0.001)
    ctx = api.Context(0); ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH); h = ctx.trace_primary(rays)
    ok = (h["id"] == want["id"]) & (want["id"] >= 0)
    rel = np.where(ok, np.abs(h["n"] - want["n"]).max(axis=1), 0)
    if os.environ.get("BY_T"):
        rel = np.where(ok, np.abs(h["t"] - want["t"]) / np.maximum(np.abs(want["t"]), 1e-30), 0)
    for i in np.argsort(rel)[-3:]:
        pid = want["id"][i]
        print(f"seed {seed} scale {scale} ray {i}: prim {pid} type {flat.prim_type[pid]} geom {flat.prim_geom[pid][:5]} xf {flat.prim_xform[pid]} t want {want['t'][i]:.9g} got {h['t'][i]:.9g} n want {want['n'][i]} got {h['n'][i]} err {rel[i]:.2e} o {rays[i][:3]} d {rays[i][3:]}")
    ctx.close()
