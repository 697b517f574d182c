"""One scene of tests/test_gpu_fuzz.py::test_random_scene again, with the rays whose closest hit
differs most from the double-precision restatement printed in full (ray, object record, both
distances, both normals).  python tests/replay/fuzz_case.py SEED [n_worst]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    seed = int(sys.argv[1])
    n_worst = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from oracle.pyoracle import Oracle
    from raytrace_amd import api, scene as S
    import test_gpu_fuzz as F
    oracle = Oracle()
    rng = np.random.default_rng(1000 + seed)
    flat, cam, scale, offset = F._random_scene(S, rng)
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 5, k, 0)[0] for k in range(1500)]
    for _ in range(1500):
        o = offset + rng.uniform(-6, 6, 3) * scale * [1, 0.3, 1] + [0, 1.5 * scale, 0]
        rays.append(np.concatenate([o, F._unit(rng.normal(size=3)) * rng.uniform(0.3, 2)]))
    rays = np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)
    want = oracle.trace(flat, rays, t_min=0.001)
    print("scale", scale, "offset", np.asarray(offset).tolist(), "primitives", flat.n_prims)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        hits = ctx.trace_primary(rays)
        ctx.close()
        same = (hits["id"] == want["id"]) & (want["id"] >= 0)
        cos = np.abs(np.sum(want["n"] * F._unit(rays[:, 3:]), axis=1))
        dn = np.where(same, np.abs(hits["n"] - want["n"]).max(axis=1), 0.0)
        dt = np.where(same, np.abs(hits["t"] - want["t"]) / np.maximum(np.abs(want["t"]), 1e-30), 0.0)
        print("accel", accel, "ids differ on", int((hits["id"] != want["id"]).sum()), "rays")
        for k in np.argsort(-dn)[:n_worst]:
            print(f" ray {k}: id {int(want['id'][k])} cos {cos[k]:.3e} |d| {np.linalg.norm(rays[k, 3:]):.4f}")
            print("   o,d     ", rays[k].tolist())
            print(f"   t gpu {float(hits['t'][k])!r} want {float(want['t'][k])!r} rel {dt[k]:.3e}")
            print("   n gpu   ", hits["n"][k].tolist())
            print("   n want  ", want["n"][k].tolist(), f"max diff {dn[k]:.3e}")
            for name in ("front_face",):
                print("   front   ", bool(hits[name][k]), bool(want["front"][k]))
            print("   record  ", F_describe(flat, int(want["id"][k])))


def F_describe(flat, obj):
    """Type, geometry and transform of primitive obj of the flattened scene."""
    x = int(flat.prim_xform[obj]) if flat.prim_xform is not None else -1
    return dict(type=int(flat.prim_type[obj]), geom=flat.prim_geom[obj].tolist(), mat=int(flat.mat_type[flat.prim_mat[obj]]),
                xform=flat.xform[x].tolist() if x >= 0 else None)


if __name__ == "__main__":
    main()
