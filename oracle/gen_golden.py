"""TEST INFRASTRUCTURE — writes the golden fixtures under tests/golden/ (and the stored default
scene under raytrace_amd/data/) by running the reference's own compiled main.cc
(oracle/_ref/libref_main.so, built by `make -C oracle ref`).

Run from the repository root, in a container that holds the reference checkout:

    python oracle/gen_golden.py

Every expected value in the fixtures is an output of the reference's unmodified functions
(see oracle/ref_harness.cc for the list with file:line); this script only chooses inputs.
Inputs that the GPU receives as float (ray origins/directions) are rounded to float32 BEFORE
the reference evaluates them, so both sides see identical numbers.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Reference, build  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "raytrace_amd", "data")

CAM_DEFAULT = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0,
                   aspect=1.5, aperture=0.1, focus_dist=10.0)          # main.cc:90,100-106


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def perp_basis(d):
    a = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    e1 = unit(np.cross(d, a))
    e2 = np.cross(d, e1)
    return e1, e2


def primary_ray_cases(ref, scene, rng):
    """Rays of five kinds; returns (rays float32-exact [n,6], kind[n])."""
    cam = ref.set_camera(**CAM_DEFAULT)
    W, H = 1200, 800
    rays, kind = [], []

    # 0: camera rays through random pixels, lens and jitter drawn from the keyed stream
    for k in range(3000):
        i, j = int(rng.integers(0, W)), int(rng.integers(0, H))
        ref.set_key(77, j * W + i, k, 0)
        s = (i + rng.random()) / (W - 1)
        t = (j + rng.random()) / (H - 1)
        rays.append(ref.get_ray(s, t)); kind.append(0)

    o = np.array(CAM_DEFAULT["lookfrom"], dtype=np.float64)
    centers, radii = scene[:, 0:3], scene[:, 3]

    # 1: grazing rays: impact parameter r*(1 +- delta) around random spheres' silhouettes
    picks = np.concatenate([[0, len(scene) - 3, len(scene) - 2, len(scene) - 1],
                            rng.choice(np.arange(1, len(scene) - 3), 220, replace=False)])
    for idx in picks:
        c, r = centers[idx], radii[idx]
        oc = c - o
        dist = np.linalg.norm(oc)
        e1, e2 = perp_basis(oc / dist)
        for delta in (1e-3, -1e-3, 2e-4, -2e-4):
            phi = rng.random() * 2 * np.pi
            b = r * (1 + delta)
            s = b / np.sqrt(1 - (b / dist) ** 2)
            q = c + s * (np.cos(phi) * e1 + np.sin(phi) * e2)
            rays.append(np.concatenate([o, q - o])); kind.append(1)

    # 2: ground-contact rays: aimed next to where small spheres touch the r=1000 ground
    for idx in rng.choice(np.arange(1, len(scene) - 3), 300, replace=False):
        c = centers[idx]
        rho = 10 ** rng.uniform(-3, -1)
        phi = rng.random() * 2 * np.pi
        q = np.array([c[0] + rho * np.cos(phi), 0.0, c[2] + rho * np.sin(phi)])
        q[1] = np.sqrt(1000.0 ** 2 - q[0] ** 2 - q[2] ** 2) - 1000.0 + rng.uniform(0, 2e-3)
        rays.append(np.concatenate([o, q - o])); kind.append(2)

    # 3: rays that start inside glass spheres
    glass = np.where(scene[:, 4] == 2)[0]
    for k in range(400):
        idx = glass[k % len(glass)]
        c, r = centers[idx], radii[idx]
        p = c + unit(rng.normal(size=3)) * r * rng.random() ** (1 / 3) * 0.98
        d = unit(rng.normal(size=3)) * rng.uniform(0.3, 3.0)
        rays.append(np.concatenate([p, d])); kind.append(3)

    rays = f32(np.array(rays))

    # 4: bounce rays: origins on the surfaces the camera rays hit, un-normalised directions
    first = ref.trace(rays[:3000])
    hit = np.where(first["id"] >= 0)[0][:900]
    p = f32(first["p"][hit])
    d = f32(first["n"][hit] + unit(rng.normal(size=(len(hit), 3))) * 0.999)
    rays = np.concatenate([rays, np.concatenate([p, d], axis=1)])
    kind += [4] * len(hit)
    return rays, np.array(kind, dtype=np.int32), cam


def grazing_margin(scene, rays, kind):
    """|b/r - 1| of each grazing ray against its nearest silhouette, in double, after rounding."""
    out = np.full(len(rays), np.inf)
    for k in np.where(kind == 1)[0]:
        o, d = rays[k, :3], unit(rays[k, 3:])
        oc = scene[:, 0:3] - o
        b = np.linalg.norm(np.cross(oc, d), axis=1)
        out[k] = np.min(np.abs(b / scene[:, 3] - 1))
    return out


def main():
    build(("ref",))
    ref = Reference()
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(DATA, exist_ok=True)
    rng = np.random.default_rng(20240607)

    # ---- the default scene (main.cc:37-85, g++ build) ------------------------------------
    scene = ref.build_default_scene()
    np.save(os.path.join(DATA, "rtiow_gxx_scene.npy"), scene)

    # ---- camera records (camera.h:19-38) ----------------------------------------------
    cam_cases = [CAM_DEFAULT,
                 dict(CAM_DEFAULT, aspect=float(np.float32(16.0 / 9.0)), vfov=35.0, aperture=0.0),
                 dict(lookfrom=(278, 278, -800), lookat=(278, 278, 0), vup=(0, 1, 0), vfov=40.0,
                      aspect=16.0 / 9.0, aperture=0.0, focus_dist=10.0),
                 dict(lookfrom=(-3, 7.5, 2), lookat=(1, 0.5, -4), vup=(0.1, 1, 0.05), vfov=63.0,
                      aspect=1.0, aperture=0.7, focus_dist=3.25)]
    cam_in = np.array([[*c["lookfrom"], *c["lookat"], *c["vup"], c["vfov"], c["aspect"],
                        c["aperture"], c["focus_dist"]] for c in cam_cases])
    cam_out = np.array([ref.set_camera(**c) for c in cam_cases])

    # ---- primary / secondary ray closest hits (sceneobject_list.h:26, sphere.h:18) ------
    rays, kind, cam_rec = primary_ray_cases(ref, scene, rng)
    tr = ref.trace(rays)
    np.savez_compressed(os.path.join(GOLD, "primary_rays.npz"),
                        rays=rays.astype(np.float32), kind=kind, id=tr["id"], t=tr["t"], p=tr["p"],
                        n=tr["n"], front=tr["front"], graze_margin=grazing_margin(scene, rays, kind),
                        t_min=0.001)

    # ---- per-function vectors -------------------------------------------------------------
    ref.set_camera(**CAM_DEFAULT)
    n = 128
    gr_in = np.zeros((n, 5))      # s, t, seed, pixel, sample  (draws start at index 2 of segment 0)
    gr_out = np.zeros((n, 6))
    for k in range(n):
        s, t = rng.random(), rng.random()
        seed, pixel, sample = int(rng.integers(1, 2 ** 31)), int(rng.integers(0, 2 ** 20)), int(rng.integers(0, 500))
        ref.lib.ref_set_key(seed, pixel, sample, 0)
        # skip the two jitter draws like the render loop does (main.cc:117-118)
        ref.lib.rt_ref_rand(); ref.lib.rt_ref_rand()
        gr_in[k] = (s, t, seed, pixel, sample)
        gr_out[k] = ref.get_ray(s, t)

    samp_in = np.zeros((n, 4), np.int64)   # seed pixel sample segment
    samp_out = {w: np.zeros((n, 3)) for w in ("in_unit_sphere", "unit_vector", "in_unit_disk")}
    samp_used = {w: np.zeros(n, np.int64) for w in samp_out}
    for k in range(n):
        key = (int(rng.integers(1, 2 ** 31)), int(rng.integers(0, 2 ** 20)), int(rng.integers(0, 500)),
               int(rng.integers(1, 50)))
        samp_in[k] = key
        for w in samp_out:
            ref.set_key(*key)
            d0 = ref.lib.ref_draw_count()
            samp_out[w][k] = ref.sample(w)
            samp_used[w][k] = ref.lib.ref_draw_count() - d0

    # scatter: hit records taken from real closest hits so that (p, normal, front) are consistent
    hitmask = tr["id"] >= 0
    cand = np.where(hitmask)[0]
    by_type = {t: [k for k in cand if scene[tr["id"][k], 4] == t] for t in (0, 1, 2)}
    sel = np.array(by_type[0][:150] + by_type[1][:150] + by_type[2][:250])
    sc_in = np.zeros((len(sel), 6 + 3 + 3 + 1 + 1 + 4))
    sc_out = np.zeros((len(sel), 10))
    for q, k in enumerate(sel):
        key = (int(rng.integers(1, 2 ** 31)), int(rng.integers(0, 2 ** 20)), int(rng.integers(0, 500)),
               int(rng.integers(1, 50)))
        ref.set_key(*key)
        ok, att, so, sd = ref.scatter(tr["id"][k], rays[k], tr["p"][k], tr["n"][k], tr["front"][k])
        sc_in[q] = (*rays[k], *tr["p"][k], *tr["n"][k], tr["front"][k], tr["id"][k], *key)
        sc_out[q] = (ok, *att, *so, *sd)

    rr_v = unit(rng.normal(size=(64, 3)))
    rr_n = unit(rng.normal(size=(64, 3)))
    rr_n = np.where((np.sum(rr_v * rr_n, axis=1) > 0)[:, None], -rr_n, rr_n)   # n opposes v
    rr_eta = np.where(rng.random(64) < 0.5, 1 / 1.5, 1.5)
    refl = np.array([ref.reflect(v, nn) for v, nn in zip(rr_v, rr_n)])
    refr = np.array([ref.refract(v, nn, e) for v, nn, e in zip(rr_v, rr_n, rr_eta)])

    wc_spp = 16
    wc_in = np.concatenate([rng.random((200, 3)) * wc_spp * 1.2, np.zeros((1, 3)),
                            np.full((1, 3), wc_spp * 0.999 ** 2), np.full((1, 3), 5.0 * wc_spp),
                            (np.arange(0, 256)[:, None] / 256.0) ** 2 * wc_spp * np.ones((1, 3))])
    wc_out = ref.write_color(wc_in, wc_spp)

    # ray_color on explicit rays at several depth limits (main.cc:10-35)
    rc_rays = rays[:256]
    rc_in = np.zeros((256, 4), np.int64)   # depth seed pixel sample
    rc_out = np.zeros((256, 3))
    rc_rays_n = np.zeros(256, np.int64)
    for k in range(256):
        depth = [0, 1, 2, 5, 50][k % 5]
        key = (int(rng.integers(1, 2 ** 31)), int(rng.integers(0, 2 ** 20)), int(rng.integers(0, 500)))
        rc_in[k] = (depth, *key)
        rc_out[k], rc_rays_n[k] = ref.ray_color(rc_rays[k], depth, *key)

    np.savez_compressed(os.path.join(GOLD, "func_vectors.npz"),
                        cam_in=cam_in, cam_out=cam_out,
                        get_ray_cam=cam_rec, get_ray_in=gr_in, get_ray_out=gr_out,
                        samp_in=samp_in,
                        **{f"samp_{w}": v for w, v in samp_out.items()},
                        **{f"samp_used_{w}": v for w, v in samp_used.items()},
                        scatter_in=sc_in, scatter_out=sc_out,
                        rr_v=rr_v, rr_n=rr_n, rr_eta=rr_eta, reflect=refl, refract=refr,
                        wc_spp=wc_spp, wc_in=wc_in, wc_out=wc_out,
                        rc_rays=rc_rays, rc_in=rc_in, rc_out=rc_out, rc_nrays=rc_rays_n)

    # ---- keyed-stream images (main.cc:111-126 through the unmodified ray_color) ---------------
    ref.set_camera(**CAM_DEFAULT)
    for (W, H, spp, seed) in ((120, 80, 16, 1), (240, 160, 64, 2), (60, 40, 1024, 3)):
        sums, nrays = ref.render_keyed(W, H, spp, 50, seed)
        p3 = ref.write_color(sums.reshape(-1, 3), spp).reshape(H, W, 3)
        np.savez_compressed(os.path.join(GOLD, f"image_{W}x{H}_{spp}spp.npz"),
                            sums=sums, rays=nrays, p3=p3.astype(np.uint8), W=W, H=H, spp=spp,
                            depth=50, seed=seed, cam=np.array([*CAM_DEFAULT["lookfrom"],
                                                               *CAM_DEFAULT["lookat"], *CAM_DEFAULT["vup"],
                                                               CAM_DEFAULT["vfov"], CAM_DEFAULT["aspect"],
                                                               CAM_DEFAULT["aperture"], CAM_DEFAULT["focus_dist"]]))
        print(f"image {W}x{H}@{spp}: rays {nrays} ({nrays / (W * H * spp):.3f}/sample)")
    # a second-half sample window of the small image: exercises sample_begin
    sums, nrays = ref.render_keyed(120, 80, 8, 50, 1, sample_begin=8)
    np.savez_compressed(os.path.join(GOLD, "image_120x80_8spp_from8.npz"), sums=sums, rays=nrays)
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
