"""One fuzz scene rendered under the BVH and under the linear scan: which pixels differ, and
from which sample and depth limit on?  usage: SEED=5899 python tools/fuzz_strategy.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _random_scene
from raytrace_amd import api, scene as S
seed = int(os.environ.get("SEED", "5899"))
rng = np.random.default_rng(1000 + seed)
flat, cam, scale, offset = _random_scene(S, rng)
W, H, spp = 72, 48, 8
ctxs = {}
for name, accel in (("bvh", api.RT_ACCEL_BVH), ("dev", api.RT_ACCEL_BVH_DEVICE), ("list", api.RT_ACCEL_LIST)):
    c = api.Context(0); c.upload_scene(flat, accel=accel); c.set_camera(**cam); ctxs[name] = c
print("scale", scale, "offset", offset, "prims", len(flat.prim_type))
for s in range(spp):
    for D in (1, 2, 3, 4, 6, 10, 50):
        imgs = {}
        rays = {}
        for name, c in ctxs.items():
            imgs[name] = c.render_host(api.Context.params(W, H, 1, D, 77 + seed, sample_begin=s)); rays[name] = c.stats()["rays"]
        for other in ("dev", "list"):
            bad = np.argwhere((imgs["bvh"] != imgs[other]).any(axis=2))
            if len(bad):
                print(f"sample {s} depth {D}: bvh vs {other}: pixels (j,i) {bad[:4].tolist()} rays {rays}")
# closest hits of many random rays under the three strategies
rng2 = np.random.default_rng(7)
N = int(os.environ.get("NRAYS", "2000000"))
o = offset + rng2.uniform(-8, 8, (N, 3)) * scale * [1, 0.4, 1] + [0, 1.5 * scale, 0]
d = rng2.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d *= rng2.uniform(0.3, 2, (N, 1))
rays = np.concatenate([o, d], axis=1).astype(np.float32).astype(np.float64)
hits = {name: c.trace_primary(rays) for name, c in ctxs.items()}
for other in ("dev", "list"):
    bad = np.where(hits["bvh"]["id"] != hits[other]["id"])[0]
    print(f"bvh vs {other}: {len(bad)} of {N} rays disagree on the object")
    for i in bad[:5]:
        print("  ray", rays[i], "bvh id", hits["bvh"]["id"][i], "t", hits["bvh"]["t"][i], other, "id", hits[other]["id"][i], "t", hits[other]["t"][i])
    same = hits["bvh"]["id"] == hits[other]["id"]
    print("  t differs on", int((hits["bvh"]["t"][same] != hits[other]["t"][same]).sum()), "of the agreeing rays")
# walk the differing path with the checker and trace its rays under both strategies
from oracle.pyoracle import Oracle
orc = Oracle("f64")
cd = orc.camera(**cam)
i, j, smp = int(os.environ.get("PI", "6")), int(os.environ.get("PJ", "16")), int(os.environ.get("PS", "2"))
sd = 77 + seed; pix = j * W + i
u0 = (orc.draw(sd, pix, smp, 0, 0) >> 8) * 2.0 ** -24; u1 = (orc.draw(sd, pix, smp, 0, 1) >> 8) * 2.0 ** -24
ray, _ = orc.get_ray(cd, (i + u0) / (W - 1), (j + u1) / (H - 1), sd, pix, smp)
for seg in range(1, 6):
    h = orc.trace(flat, [ray], t_min=0.001)
    g = {name: c.trace_primary(np.asarray([ray], np.float64)) for name, c in ctxs.items()}
    print(f"ray {seg}: o {ray[:3]} d {ray[3:]} checker id {h['id'][0]} t {h['t'][0]:.9g}; " + "; ".join(f"{n} id {g[n]['id'][0]} t {g[n]['t'][0]:.9g}" for n in g))
    pid = int(h["id"][0])
    if pid < 0: break
    print("   prim", pid, "type", flat.prim_type[pid], "geom", flat.prim_geom[pid], "mat", flat.prim_mat[pid], "xform", flat.prim_xform[pid])
    ok, a, b, c3, e = orc.scatter(flat, flat.prim_mat[pid], ray, h["p"][0], h["n"][0], h["front"][0], h["uv"][0], sd, pix, smp, seg)
    print("   scatter ok", ok, a, b, c3, e)
    if not ok: break
    ray = np.concatenate([b, c3])
for D in (2, 50):
    for name, c in ctxs.items():
        img = c.render_host(api.Context.params(W, H, 1, D, 77 + seed, sample_begin=smp))
        print("depth", D, name, "pixel value", [float.hex(float(v)) for v in img[j, i, :3]], img[j, i, :3])
    so, _ = orc.render(flat, cd, W, H, 1, D, 77 + seed, sample_begin=smp, window=(i, i + 1, j, j + 1))
    print("depth", D, "checker", so[j, i])
