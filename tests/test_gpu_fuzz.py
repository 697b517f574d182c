"""Differential test on seeded random scenes: spheres of every material (also hollow and wide
ones), rects, boxes under rotate_y / translate wrappers, textures, either background; closest
hits and small renders against the double-precision restatement, under every closest-hit
strategy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _random_scene(S, rng):
    tex = S.image_texture(32, 16, S.procedural_image(32, 16, seed=int(rng.integers(1, 40))))
    chk = S.checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))

    def material():
        k = rng.integers(0, 7)
        if k == 0:
            return S.lambertian(rng.uniform(0.1, 0.9, 3))
        if k == 1:
            return S.metal(rng.uniform(0.5, 1.0, 3), float(rng.choice([0.0, 0.3, 1.5])))
        if k == 2:
            return S.dielectric(float(rng.choice([1.5, 1.33, 2.4])))
        if k == 3:
            return S.lambertian(tex)
        if k == 4:
            return S.lambertian(chk)
        if k == 5:
            return S.diffuse_light(rng.uniform(1, 5, 3))
        return S.lambertian_bg(tex)

    scale = float(rng.choice([1.0, 30.0, 400.0]))             # scene size: coordinates up to ~10 * scale
    offset = rng.uniform(-5, 5, 3) * scale * float(rng.choice([0.0, 1.0]))
    w = S.sceneobject_list()
    if rng.random() < 0.7:
        w.add(S.sphere(offset + [0, -1000 * scale, 0], 1000 * scale, S.lambertian(rng.uniform(0.3, 0.7, 3))))   # a wide sphere as ground
    for _ in range(int(rng.integers(1, 60))):
        c = offset + rng.uniform(-6, 6, 3) * scale * [1, 0.3, 1] + [0, 1.5 * scale, 0]
        r = float(rng.uniform(0.2, 1.2)) * scale
        m = material()
        obj = S.sphere(c, r, m)
        if isinstance(m, S.dielectric) and rng.random() < 0.5:
            w.add(obj)
            obj = S.sphere(c, -0.85 * r, m)                    # hollow glass: the reference's negative-radius trick
        if rng.random() < 0.2:
            obj = S.translate(S.rotate_y(S.sphere(c - offset, r, m), float(rng.uniform(-180, 180))), offset)
        w.add(obj)
    for _ in range(int(rng.integers(0, 4))):
        a0, b0 = rng.uniform(-6, 2, 2) * scale
        rect = rng.choice([S.xy_rect, S.xz_rect, S.yz_rect])(a0, a0 + rng.uniform(1, 5) * scale, b0, b0 + rng.uniform(1, 5) * scale,
                                                             float(rng.uniform(-5, 5)) * scale, material())
        w.add(S.translate(rect, offset))
    for _ in range(int(rng.integers(0, 3))):
        p0 = rng.uniform(-4, 2, 3) * scale
        bx = S.box(p0, p0 + rng.uniform(0.5, 2.5, 3) * scale, material())
        w.add(S.translate(S.rotate_y(bx, float(rng.uniform(-90, 90))), offset + rng.uniform(-2, 2, 3) * scale))
    bg = int(rng.choice([S.RT_BG_SKY, S.RT_BG_SOLID]))
    flat = S.flatten(w, background=bg, bg_color=(0.1, 0.1, 0.15))
    eye = offset + np.array([0.0, 3.0, 14.0]) * scale
    cam = dict(lookfrom=tuple(eye), lookat=tuple(offset + [0, 1.0 * scale, 0]), vup=(0, 1, 0), vfov=40.0, aspect=1.5,
               aperture=float(rng.choice([0.0, 0.2])) * scale, focus_dist=14.0 * scale)
    return flat, cam, scale, offset


# scenes that once disagreed, kept whatever the count: 5899 holds a ray that meets a sphere and a
# rect at exactly the same single-precision distance (the scan took the sphere, the tree walk the
# rect, which comes later in the list and is the one that wins a tie); 2161 has the longest glass
# paths met so far; 14114 sees a sphere of radius 210 from 34 radii away at a grazing angle, where
# |unit direction|^2 - 1 (1e-7) times the squared distance is as large as the discriminant
# 18788 has a checker texture that changes sign across a rect's plane within an ulp of the plane's
# coordinate
_KEPT_SEEDS = [2161, 5899, 14114, 18788, 56104, 343970, 345337, 545253]   # 56104: a root 6.6e-7 above t_min; 343970: one 1 % above it, 2000 from the origin; 345337: no ray meets anything; 545253: a normal at cos 1.6e-3
_N_SEEDS = int(__import__("os").environ.get("RT_FUZZ_SEEDS", "40"))
_FIRST_SEED = int(__import__("os").environ.get("RT_FUZZ_FIRST", "0"))          # long runs by hand, in parts


@pytest.mark.parametrize("seed", list(range(_FIRST_SEED, _FIRST_SEED + _N_SEEDS)) +
                         [k for k in _KEPT_SEEDS if not _FIRST_SEED <= k < _FIRST_SEED + _N_SEEDS])
def test_random_scene(seed, oracle):
    from raytrace_amd import api, scene as S
    rng = np.random.default_rng(1000 + seed)
    flat, cam, scale, offset = _random_scene(S, rng)
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 5, k, 0)[0] for k in range(1500)]
    for _ in range(1500):                                      # from inside the scene, any direction
        o = offset + rng.uniform(-6, 6, 3) * scale * [1, 0.3, 1] + [0, 1.5 * scale, 0]
        rays.append(np.concatenate([o, _unit(rng.normal(size=3)) * rng.uniform(0.3, 2)]))
    rays = np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)
    want = oracle.trace(flat, rays, t_min=0.001)
    W, H, spp = 72, 48, 8
    sums, n_rays = oracle.render(flat, cd, W, H, spp, 50, 77 + seed, threads=8)
    first = None
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE, api.RT_ACCEL_LIST):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        hits = ctx.trace_primary(rays)
        bad = hits["id"] != want["id"]
        # a root that close to t_min is the other kind of ray single precision cannot decide like
        # double: the sphere records are rounded to float, which moves a surface by 1e-7 of its
        # coordinates (1e-4 in a scene that stands 2000 from the origin, a tenth of the distance t_min
        # stands for); the same sphere is then hit at its far side instead.  Counted with the
        # silhouette rays.  (Scene 343970 of the generator: a root 1 % above t_min, 2000 from the origin.)
        reach = np.abs(rays[:, :3]).max(axis=1) + 4.0 * scale
        near_t_min = 0.001 * 1.001 + 5e-7 * reach / np.linalg.norm(rays[:, 3:], axis=1)
        near = (np.minimum(hits["t"], want["t"]) < near_t_min) & (want["id"] >= 0)
        bad |= near & ~(np.abs(hits["t"] - want["t"]) <= 1e-5 * np.abs(want["t"]))     # (a root that close AND decided the other way)
        assert bad.sum() <= 3, (accel, int(bad.sum()), np.where(bad)[0][:5], hits["id"][bad][:5], want["id"][bad][:5])
        # a ray that grazes its sphere closer than cos = 1e-3 has no well-defined t in single
        # precision (the golden rays keep clear of silhouettes for the same reason)
        cos = np.abs(np.sum(want["n"] * _unit(rays[:, 3:]), axis=1))
        ok = ~bad & (want["id"] >= 0) & (cos > 1e-3)
        if ok.any():                                               # (a scene none of the rays meets: nothing to compare)
            rel = np.abs(hits["t"][ok] - want["t"][ok]) / np.abs(want["t"][ok])
            worst = np.where(ok)[0][int(np.argmax(rel))]
            assert rel.max() <= 1e-5, (accel, float(rel.max()), int(worst), float(hits["t"][worst]), float(want["t"][worst]),
                                       int(want["id"][worst]), rays[worst].tolist(), float(scale), np.asarray(offset).tolist())
            # a normal moves by (shift of the hit point) / radius, and the hit point of a grazing ray by
            # (shift of the surface: 1e-7 of its coordinates, the rounding of the records) / cos
            # (scene 545253 of the generator: cos 1.6e-3, t off by 1.4e-6 of itself, the normal by 6.7e-5)
            # that term only where a ray grazes (cos < 1e-2: up to 2e-4 at the 1e-3 the rays are kept above); elsewhere 1e-5
            assert (np.abs(hits["n"][ok] - want["n"][ok]).max(axis=1) <= 1e-5 + np.where(cos[ok] < 1e-2, 2e-7 / cos[ok], 0.0)).all()
            assert np.array_equal(hits["front_face"][ok], want["front"][ok])
        img = ctx.render_host(api.Context.params(W, H, spp, 50, 77 + seed))
        st = ctx.stats()
        if accel != api.RT_ACCEL_LIST:
            # the same tree read from HBM instead of LDS, and walked in time slices (what scenes too
            # large for LDS and trees deeper than 16 levels get): the same frame, the same ray count
            # (and with two entries of each lane's stack in LDS, the rest in device memory)
            # (the last number: the share of unfinished lanes at which a wave leaves a time-sliced walk, in 128ths; -1 = never for that)
            for flags, slice_steps, entries, share in ((api.RT_FLAG_NO_LDS_SCENE, 0, 0, 0), (0, 3, 0, -1), (api.RT_FLAG_NO_LDS_SCENE, 5, 0, -1),
                                                        (api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES, 0, 0, 0), (api.RT_FLAG_NO_LDS_SCENE, 7, 2, 0),
                                                        (0, 40, 0, 96), (api.RT_FLAG_NO_LDS_SCENE, 200, 0, 32),
                                                        # (whole walks; scenes of spheres only: walks left between rounds of leaves)
                                                        (0, -1, 0, 0), (0, 0, 0, 128), (api.RT_FLAG_NO_LDS_SCENE, 0, 0, 100)):
                other = ctx.render_host(api.Context.params(W, H, spp, 50, 77 + seed, flags=flags, traversal_slice=slice_steps, stack_lds_entries=entries,
                                                           traversal_exit_share=share))
                assert np.array_equal(other, img), (accel, flags, slice_steps, entries, share)
                assert ctx.stats()["rays"] == st["rays"]
        if accel == api.RT_ACCEL_BVH:
            # three interleaved shards of 4-row tiles, another tile shape: the same bits
            whole = np.zeros_like(img)
            for g in range(3):
                part = ctx.render_host(api.Context.params(W, H, spp, 50, 77 + seed, tiles=(4, g, 3), lanes_per_pixel=16))
                rows = [r for t in range(g, H // 4, 3) for r in range(4 * t, 4 * t + 4)]
                whole[rows] = part
            assert np.array_equal(whole, img)
        ctx.close()
        if first is None:
            first = img
        else:
            assert np.array_equal(img, first), accel            # the frame does not depend on the strategy
        assert np.isfinite(img).all()
        # same keyed paths: float-vs-double branch flips are rare, so the images are close and
        # the number of rays traced nearly equal.  Over 1500 scenes the difference is centred on
        # zero (298 positive, 303 negative, 899 equal; relative sd 1.8e-4, largest 1.5e-3); the
        # tail comes from paths of 30-50 bounces inside glass balls, where one flipped
        # reflect-or-refract draw changes the rest of the path (one scene in 4000 reached 2.6e-3)
        assert abs(st["rays"] - n_rays) <= max(20, 4e-3 * n_rays), (accel, st["rays"], n_rays)
        peak = max(1.0, float(sums.max() / spp))
        assert np.sqrt(np.mean((img[..., :3] / spp - sums / spp) ** 2)) <= 0.02 * peak


def _random_field(S, rng, plain=False):
    """Hundreds to thousands of small spheres: trees of 8 to 14 levels, in LDS or beyond it.
    plain: no emitter among the materials, so that the frame takes the kernels built for scenes of plainly
    coloured spheres (the walk left between rounds of leaves, the draws of a trip in one place)."""
    n = int(rng.choice([150, 500, 1500, 4000]))
    scale = float(rng.choice([1.0, 50.0]))
    offset = rng.uniform(-3, 3, 3) * scale * float(rng.choice([0.0, 1.0]))
    side = 3.0 * n ** (1 / 3)
    w = S.sceneobject_list()
    w.add(S.sphere(offset + [0, -1000 * scale, 0], 1000 * scale, S.lambertian(rng.uniform(0.3, 0.7, 3))))
    mats = [S.lambertian(rng.uniform(0.1, 0.9, 3)) for _ in range(6)] + [S.metal(rng.uniform(0.5, 1, 3), 0.0), S.metal(rng.uniform(0.5, 1, 3), 0.4),
                                                                         S.dielectric(1.5), S.diffuse_light((3, 3, 3))]
    if plain:
        mats[-1] = mats[0]                                      # (after the draws, so that a seed's field is the same but for its lights)
    c = rng.uniform(-0.5, 0.5, (n, 3)) * [side, 0.15 * side, side] + [0, 0.12 * side, 0]
    r = rng.uniform(0.15, 0.6, n)
    for k in range(n):
        w.add(S.sphere(offset + c[k] * scale, float(r[k]) * scale, mats[int(rng.integers(0, len(mats)))]))
    flat = S.flatten(w, background=int(rng.choice([S.RT_BG_SKY, S.RT_BG_SOLID])), bg_color=(0.1, 0.1, 0.15))
    inside = rng.random() < 0.3                                 # a camera among the spheres now and then
    eye = offset + (np.array([0.1, 0.08, 0.05]) if inside else np.array([0.0, 0.5, 1.1])) * side * scale
    cam = dict(lookfrom=tuple(eye), lookat=tuple(offset + [0, 0.05 * side * scale, 0]), vup=(0, 1, 0), vfov=float(rng.choice([20.0, 50.0, 90.0])),
               aspect=1.5, aperture=float(rng.choice([0.0, 0.05])) * scale, focus_dist=side * scale)
    return flat, cam, scale, offset, side


_FIRST_FIELD = int(__import__("os").environ.get("RT_FUZZ_FIELD_FIRST", "0"))
_N_FIELDS = int(__import__("os").environ.get("RT_FUZZ_FIELDS", "6"))
# 84596: a camera among 500 spheres at a field of view of 20 degrees, emitters in sight: 9.8 % of the pixels differ
# from the double-precision frame at depth limit 50 (none up to 8, 0.04 % at 12), the largest image difference met
_KEPT_FIELDS = [84596]


@pytest.mark.parametrize("seed", list(range(_FIRST_FIELD, _FIRST_FIELD + _N_FIELDS)) +
                         [k for k in _KEPT_FIELDS if not _FIRST_FIELD <= k < _FIRST_FIELD + _N_FIELDS])
def test_random_sphere_field(seed, oracle):
    from raytrace_amd import api, scene as S
    rng = np.random.default_rng(50000 + seed)
    flat, cam, scale, offset, side = _random_field(S, rng, plain=seed % 3 == 1)      # every third field without lights
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 5, k, 0)[0] for k in range(1000)]
    for _ in range(1000):
        o = offset + rng.uniform(-0.5, 0.5, 3) * side * scale * [1, 0.2, 1] + [0, 0.12 * side * scale, 0]
        rays.append(np.concatenate([o, _unit(rng.normal(size=3)) * rng.uniform(0.3, 2)]))
    # rays along an axis, in an axis plane, and a hair off either: the slab test's special cases
    # (a reciprocal of +-1e30, per-axis rounding margins that grow with |o/d|)
    for k in range(600):
        o = offset + rng.uniform(-0.5, 0.5, 3) * side * scale * [1, 0.2, 1] + [0, 0.12 * side * scale, 0]
        dvec = _unit(rng.normal(size=3)) * rng.uniform(0.3, 2)
        zero = rng.choice(3, size=int(rng.integers(1, 3)), replace=False)
        dvec[zero] = 0.0 if k % 2 == 0 else rng.choice([-1, 1], size=len(zero)) * 10.0 ** rng.uniform(-12, -5, size=len(zero))
        rays.append(np.concatenate([o, dvec]))
    rays = np.asarray(rays, np.float64).astype(np.float32).astype(np.float64)
    want = oracle.trace(flat, rays, t_min=0.001)
    # depth limits: 3 and 8, where single precision decides every branch of a path like the double-precision
    # checker (tests/replay/field_case.py: no pixel differs up to 8) and the strict bounds hold, and 50
    W, H, spp, depth = 60, 40, 4, (50 if seed in _KEPT_FIELDS else int(rng.choice([3, 8, 50])))
    sums, n_rays = oracle.render(flat, cd, W, H, spp, depth, 9 + seed, threads=8)
    first = None
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE, api.RT_ACCEL_LIST):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        hits = ctx.trace_primary(rays)
        # the walk a frame of this scene takes (the tree in LDS where it fits) and the single-precision nodes: the same bits
        assert ctx.trace_primary(rays, walk=0).tobytes() == hits.tobytes(), accel
        assert ctx.trace_primary(rays, walk=api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES).tobytes() == hits.tobytes(), accel
        bad = hits["id"] != want["id"]
        # a root that close to t_min is the other kind of ray single precision cannot decide like
        # double: the sphere records are rounded to float, which moves a surface by 1e-7 of its
        # coordinates (1e-4 in a scene that stands 2000 from the origin, a tenth of the distance t_min
        # stands for); the same sphere is then hit at its far side instead.  Counted with the
        # silhouette rays.  (Scene 343970 of the generator: a root 1 % above t_min, 2000 from the origin.)
        reach = np.abs(rays[:, :3]).max(axis=1) + 4.0 * scale
        near_t_min = 0.001 * 1.001 + 5e-7 * reach / np.linalg.norm(rays[:, 3:], axis=1)
        near = (np.minimum(hits["t"], want["t"]) < near_t_min) & (want["id"] >= 0)
        bad |= near & ~(np.abs(hits["t"] - want["t"]) <= 1e-5 * np.abs(want["t"]))     # (a root that close AND decided the other way)
        assert bad.sum() <= 3, (accel, int(bad.sum()), np.where(bad)[0][:5], hits["id"][bad][:5], want["id"][bad][:5])
        cos = np.abs(np.sum(want["n"] * _unit(rays[:, 3:]), axis=1))
        ok = ~bad & (want["id"] >= 0) & (cos > 1e-3)
        rel = np.abs(hits["t"][ok] - want["t"][ok]) / np.abs(want["t"][ok])
        assert rel.max() <= 1e-5, (accel, float(rel.max()))
        # (grazing rays, cos < 1e-2 only: as in test_random_scene)
        assert (np.abs(hits["n"][ok] - want["n"][ok]).max(axis=1) <= 1e-5 + np.where(cos[ok] < 1e-2, 2e-7 / cos[ok], 0.0)).all()
        img = ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed))
        st = ctx.stats()
        if accel == api.RT_ACCEL_BVH:
            assert np.array_equal(ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, flags=api.RT_FLAG_NO_LDS_SCENE)), img)
            # time-sliced, with three entries of each lane's stack in LDS and the rest in device memory
            assert np.array_equal(ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, flags=api.RT_FLAG_NO_LDS_SCENE, traversal_slice=24,
                                                                      stack_lds_entries=3)), img)
            assert np.array_equal(ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, flags=api.RT_FLAG_NO_LDS_SCENE, traversal_slice=24,
                                                                      traversal_exit_share=-1)), img)
            assert np.array_equal(ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, traversal_slice=500, traversal_exit_share=112)), img)
            # the walk that is never left early, and the one left between rounds of leaves at two other shares
            for flags, sl, share in ((0, -1, 0), (api.RT_FLAG_NO_LDS_SCENE, -1, 0), (0, 0, 128), (api.RT_FLAG_NO_LDS_SCENE, 0, 8), (0, 0, -1)):
                assert np.array_equal(ctx.render_host(api.Context.params(W, H, spp, depth, 9 + seed, flags=flags, traversal_slice=sl,
                                                                          traversal_exit_share=share)), img), (flags, sl, share)
        ctx.close()
        if first is None:
            first = img
        else:
            assert np.array_equal(img, first), accel
        # a dense field of small mirrors and glass balls is a chaotic billiard: every bounce off a
        # convex sphere multiplies a direction error by (distance / radius), so single and double
        # precision paths part after seven or eight specular bounces (measured: no pixel differs up
        # to depth limit 2, 6 % differ at 50; ray counts over 400 scenes centred on zero with a
        # relative sd of 1e-3 at 27 648 samples, largest 4.9e-3; this test's 9 600 samples scatter twice as much)
        assert abs(st["rays"] - n_rays) <= max(40, (4e-3 if depth <= 8 else 3e-2) * n_rays), (accel, depth, st["rays"], n_rays)
        peak = max(1.0, float(sums.max() / spp))
        # (at depth limit 50 one scene in 37 000 reached 0.066 of the peak, the kept one: tests/replay/field_case.py shows
        # it by depth limit; the wider bound is that scene's alone)
        bound = 0.02 if depth <= 8 else (0.09 if seed in _KEPT_FIELDS else 0.06)
        assert np.sqrt(np.mean((img[..., :3] / spp - sums / spp) ** 2)) <= bound * peak, (accel, depth)
