"""GPU parity on the CUDA-tree features (rects, emitters, textures, BVH over many spheres) and on
edge cases.  The checker is the double-precision CPU restatement (linear scan); for these
features it follows the CUDA tree's text and is parity-unpinned beyond review."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from raytrace_amd import api as A
    return A


@pytest.fixture(scope="module")
def S():
    from raytrace_amd import scene
    return scene


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _f32(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def _check_hits(hits, want, t_rel=1e-5, n_abs=1e-5, uv_abs=None, allow_id_mismatch=0):
    bad = np.where(hits["id"] != want["id"])[0]
    assert len(bad) <= allow_id_mismatch, (len(bad), bad[:8], hits["id"][bad][:8], want["id"][bad][:8])
    ok = (want["id"] >= 0) & (hits["id"] == want["id"])
    assert (np.abs(hits["t"][ok] - want["t"][ok]) <= t_rel * np.abs(want["t"][ok])).all()
    assert np.abs(hits["n"][ok] - want["n"][ok]).max() <= n_abs
    assert np.array_equal(hits["front_face"][ok], want["front"][ok])
    if uv_abs is not None:
        # the longitude u of a sphere is undefined at its poles: skip the polar caps
        m = ok & (np.abs(want["n"][:, 1]) < 0.999)
        assert np.abs(np.stack([hits["u"][m], hits["v"][m]], 1) - want["uv"][m]).max() <= uv_abs


def _rmse(img, sums, spp):
    return float(np.sqrt(np.mean((img[..., :3].astype(np.float64) / spp - sums / spp) ** 2)))


# ---- Cornell box: rects + emitter + glass/metal spheres, black background ---------------------

def _cornell(api, S, oracle=None):
    flat = S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    cam = S.cornell_camera(1.5).desc()
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    return flat, cam, ctx


def test_cornell_closest_hits(api, S, oracle):
    flat, cam, ctx = _cornell(api, S, oracle)
    rng = np.random.default_rng(5)
    cd = oracle.camera(**cam)
    rays = []
    for k in range(3000):
        r, _ = oracle.get_ray(cd, rng.random(), rng.random(), 3, k, 0)
        rays.append(r)
    # secondary rays from inside the room, any direction (hits walls, light, spheres from inside)
    for k in range(3000):
        o = rng.uniform(5, 550, 3)
        rays.append(np.concatenate([o, _unit(rng.normal(size=3)) * rng.uniform(0.2, 2)]))
    rays = _f32(np.array(rays))
    want = oracle.trace(flat, rays)
    hits = ctx.trace_primary(rays)
    # rect uv is formed from the float hit point: |p| ~ 555 gives ~6e-5/555 relative
    _check_hits(hits, want, uv_abs=2e-6)
    assert set(np.unique(want["id"])) >= {0, 1, 2, 3, 4, 5, 6, 7}
    ctx.close()


def test_cornell_image_matches_restatement(api, S, oracle):
    flat, cam, ctx = _cornell(api, S, oracle)
    W, H, spp = 96, 64, 32
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 21))
    st = ctx.stats()
    sums, rays = oracle.render(flat, oracle.camera(**cam), W, H, spp, 50, 21, threads=8)
    # paths end on the light (emission 7) or in black: high-variance estimator, but the SAME
    # stream on both sides, so only float-vs-double branch flips differ
    assert _rmse(img, sums, spp) <= 0.03, _rmse(img, sums, spp)
    assert abs(img[..., :3].mean() / spp - sums.mean() / spp) < 5e-3
    assert abs(st["rays"] - rays) / rays < 0.005
    assert img[..., :3].max() > 0          # the light is seen
    ctx.close()


# ---- instances: boxes of six rects turned about y and moved (cuda/sceneobject.h:31-109) ---------------

def _cornell_boxes(api, S):
    flat = S.flatten(S.cornell_boxes_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    cam = S.cornell_camera(1.5).desc()
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    return flat, cam, ctx


def test_instances_closest_hits(api, S, oracle):
    flat, cam, ctx = _cornell_boxes(api, S)
    rng = np.random.default_rng(15)
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 3, k, 0)[0] for k in range(3000)]
    for k in range(3000):
        o = rng.uniform(5, 550, 3)
        rays.append(np.concatenate([o, _unit(rng.normal(size=3)) * rng.uniform(0.2, 2)]))
    # from inside the glass sphere that rides on the turned short box
    c, s = np.cos(np.deg2rad(-18.0)), np.sin(np.deg2rad(-18.0))
    glass = np.array([c * 82.5 + s * 82.5 + 130, 215.0, -s * 82.5 + c * 82.5 + 65])
    for k in range(500):
        rays.append(np.concatenate([glass + rng.uniform(-20, 20, 3), _unit(rng.normal(size=3))]))
    rays = _f32(np.array(rays))
    want = oracle.trace(flat, rays)
    hits = ctx.trace_primary(rays)
    # a ray that runs into a box edge-on may take either face in float
    _check_hits(hits, want, uv_abs=5e-6, allow_id_mismatch=2)
    assert set(range(6, 19)) <= set(np.unique(want["id"]))            # all twelve faces and the sphere
    inside = want["id"][-500:] == 18
    assert inside.sum() > 400 and (hits["front_face"][-500:][inside] == 0).all()
    ctx.close()


def test_instances_image_matches_restatement(api, S, oracle):
    flat, cam, ctx = _cornell_boxes(api, S)
    W, H, spp = 96, 64, 32
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 23))
    st = ctx.stats()
    sums, rays = oracle.render(flat, oracle.camera(**cam), W, H, spp, 50, 23, threads=8)
    assert _rmse(img, sums, spp) <= 0.03, _rmse(img, sums, spp)
    assert abs(img[..., :3].mean() / spp - sums.mean() / spp) < 5e-3
    assert abs(st["rays"] - rays) / rays < 0.005
    ctx.close()


def test_textured_sphere_instance_turns_its_texture(api, S, oracle):
    """A globe turned about y: the hit is the moved sphere's, the texture coordinates turn with
    it (the frames of a spinning planet differ only in this angle)."""
    tex = S.image_texture(64, 32, S.procedural_image(64, 32))
    rays = None
    for angle in (0.0, 75.0, -200.0):
        w = S.sceneobject_list()
        w.add(S.sphere((0, -1000, 0), 1000, S.lambertian((0.5, 0.5, 0.5))))
        w.add(S.translate(S.rotate_y(S.sphere((0, 0, 0), 1.0, S.lambertian(tex)), angle), (0.5, 1, -0.25)))
        w.add(S.translate(S.rotate_y(S.sphere((0, 0, 0), 100.0, S.lambertian_bg(tex)), angle), (300, 100, 0)))
        flat = S.flatten(w)
        cam = dict(lookfrom=(0, 2, 7), lookat=(0.5, 1, 0), vup=(0, 1, 0), vfov=30.0, aspect=1.5, aperture=0.0, focus_dist=7.0)
        ctx = api.Context(0)
        ctx.upload_scene(flat)
        ctx.set_camera(**cam)
        if rays is None:
            rng = np.random.default_rng(9)
            cd = oracle.camera(**cam)
            rays = [oracle.get_ray(cd, rng.random(), rng.random(), 4, k, 0)[0] for k in range(2000)]
            rays += [np.concatenate([[0, 2, 7], _unit(np.array([300, 100, 0]) - [0, 2, 7] + 60 * rng.normal(size=3))]) for _ in range(1000)]
            rays = _f32(np.array(rays))
        want = oracle.trace(flat, rays)
        hits = ctx.trace_primary(rays)
        _check_hits(hits, want, uv_abs=2e-5)
        assert (want["id"] == 1).sum() > 200 and (want["id"] == 2).sum() > 300
        img = ctx.render_host(api.Context.params(48, 32, 16, 10, 3))
        sums, _ = oracle.render(flat, oracle.camera(**cam), 48, 32, 16, 10, 3, threads=8)
        assert _rmse(img, sums, 16) <= 0.02
        ctx.close()


# ---- the reference GPU driver's own world (cuda/main.cu:39-189) ------------------------------------------

def test_skybox_world_of_the_reference_driver(api, S, oracle):
    flat = S.flatten(S.skybox_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    assert flat.n_prims == 6 + 100 + 6 + 125 + 7
    cam = S.skybox_camera(1.5).desc()
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    rng = np.random.default_rng(21)
    cd = oracle.camera(**cam)
    rays = [oracle.get_ray(cd, rng.random(), rng.random(), 3, k, 0)[0] for k in range(4000)]
    for k in range(2000):
        rays.append(np.concatenate([rng.uniform(20, 530, 3), _unit(rng.normal(size=3))]))
    rays = _f32(np.array(rays))
    want = oracle.trace(flat, rays)
    hits = ctx.trace_primary(rays)
    _check_hits(hits, want, uv_abs=2e-5, allow_id_mismatch=2)
    assert len(np.unique(want["id"])) > 100
    W, H, spp = 96, 64, 32
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 27))
    sums, rays_n = oracle.render(flat, cd, W, H, spp, 50, 27, threads=8)
    assert _rmse(img, sums, spp) <= 0.03, _rmse(img, sums, spp)
    assert abs(ctx.stats()["rays"] - rays_n) / rays_n < 0.005
    assert ctx.stats()["accel"] == api.RT_ACCEL_BVH
    ctx.close()


# ---- textures: checker (nested) and image, on spheres and rects -------------------------------------

def _textured_world(S):
    img = S.procedural_image(64, 32)
    tex_img = S.image_texture(64, 32, img)
    chk = S.checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    nested = S.checker_texture(chk, tex_img)
    w = S.sceneobject_list()
    w.add(S.sphere((0, -1000, 0), 1000, S.lambertian(chk)))
    w.add(S.sphere((0, 1, 0), 1.0, S.lambertian(tex_img)))
    w.add(S.sphere((-2.2, 1, 0.3), 1.0, S.lambertian(nested)))
    w.add(S.sphere((2.2, 1, -0.3), 1.0, S.metal((0.8, 0.8, 0.8), 0.1)))
    w.add(S.xy_rect(-3, 3, 0, 2.5, -2.0, S.lambertian(tex_img)))
    w.add(S.yz_rect(0, 2, -1.5, 1.5, 4.0, S.diffuse_light(chk)))
    w.add(S.xz_rect(-1, 1, -1, 1, 3.5, S.lambertian_bg(tex_img)))
    return w


def test_textured_scene(api, S, oracle):
    flat = S.flatten(_textured_world(S))
    cam = dict(lookfrom=(0, 2.5, 9), lookat=(0, 1, 0), vup=(0, 1, 0), vfov=35.0, aspect=1.5,
               aperture=0.05, focus_dist=9.0)
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    rng = np.random.default_rng(8)
    cd = oracle.camera(**cam)
    rays = _f32(np.array([oracle.get_ray(cd, rng.random(), rng.random(), 4, k, 0)[0] for k in range(4000)]))
    want = oracle.trace(flat, rays)
    hits = ctx.trace_primary(rays)
    _check_hits(hits, want, uv_abs=2e-5)
    W, H, spp = 96, 64, 32
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 33))
    sums, rays_n = oracle.render(flat, cd, W, H, spp, 50, 33, threads=8)
    # texel and checker-cell lookups flip where float and double disagree on a cell edge
    assert _rmse(img, sums, spp) <= 0.02, _rmse(img, sums, spp)
    assert abs(ctx.stats()["rays"] - rays_n) / rays_n < 0.005
    ctx.close()


def test_image_texture_texel_addressing(api, S, oracle):
    """One emissive image rect facing the camera, depth 1: every pixel is a single texel fetch."""
    img = S.procedural_image(16, 8)
    w = S.sceneobject_list()
    w.add(S.xy_rect(-2, 2, -1, 1, 0.0, S.lambertian_bg(S.image_texture(16, 8, img))))
    flat = S.flatten(w, background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    cam = dict(lookfrom=(0, 0, 5), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=30.0, aspect=2.0, aperture=0.0, focus_dist=5.0)
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    W, H, spp = 64, 32, 4
    out = ctx.render_host(api.Context.params(W, H, spp, 1, 2))
    sums, _ = oracle.render(flat, oracle.camera(**cam), W, H, spp, 1, 2, threads=4)
    diff = np.abs(out[..., :3] - sums)
    assert (diff > 1e-4).mean() < 0.01          # only texel-edge samples may differ
    assert out[..., :3].max() > 0
    ctx.close()


# ---- BVH over many spheres against the linear scan ----------------------------------------------------

def test_bvh_equals_linear_scan_20k_spheres(api, S, oracle):
    rng = np.random.default_rng(11)
    n = 20000
    w = S.sceneobject_list()
    mats = [S.lambertian((0.5, 0.5, 0.5)), S.metal((0.8, 0.8, 0.8), 0.2), S.dielectric(1.5)]
    c = np.stack([rng.uniform(-60, 60, n), rng.uniform(0, 12, n), rng.uniform(-60, 60, n)], 1)
    r = rng.uniform(0.1, 0.5, n)
    for k in range(n):
        w.add(S.sphere(c[k], r[k], mats[k % 3]))
    flat = S.flatten(w)
    rays = np.concatenate([np.tile([0, 30, -90], (3000, 1)) + rng.normal(size=(3000, 3)),
                           _unit(np.tile([0, -0.25, 1], (3000, 1)) + 0.35 * rng.normal(size=(3000, 3)))], 1)
    inside = np.concatenate([c[:1500] + rng.normal(size=(1500, 3)) * 0.6, _unit(rng.normal(size=(1500, 3)))], 1)
    rays = _f32(np.concatenate([rays, inside]))
    want = oracle.trace(flat, rays)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        hits = ctx.trace_primary(rays)
        # random overlapping spheres: a ray may graze a silhouette closer than float resolves
        _check_hits(hits, want, allow_id_mismatch=2)
        if accel == api.RT_ACCEL_BVH:
            ctx.set_camera((0, 30, -90), (0, 5, 0), (0, 1, 0), 40.0, 1.5, 0.0, 10.0)
            ctx.render_host(api.Context.params(64, 48, 2, 8, 5))
            st = ctx.stats()
            assert st["accel"] == api.RT_ACCEL_BVH and st["bvh_depth"] > 8 and st["bvh_nodes"] > 4000
        ctx.close()
    assert (want["id"] >= 0).mean() > 0.3


# ---- edge cases ---------------------------------------------------------------------------------------------

def test_edge_cases(api, S, oracle):
    cam = dict(lookfrom=(0, 0, 5), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=40.0, aspect=1.0, aperture=0.0, focus_dist=5.0)
    ctx = api.Context(0)
    # empty world: every ray sees the sky gradient (main.cc:30-34)
    empty = S.flatten(S.sceneobject_list())
    ctx.upload_scene(empty)
    ctx.set_camera(**cam)
    img = ctx.render_host(api.Context.params(32, 32, 2, 50, 1))
    sums, rays = oracle.render(empty, oracle.camera(**cam), 32, 32, 2, 50, 1)
    np.testing.assert_allclose(img[..., :3], sums, rtol=1e-5)
    assert ctx.stats()["rays"] == rays == 32 * 32 * 2
    assert (ctx.trace_primary(np.array([[0, 0, 5, 0, 0, -1]], np.float32))["id"] == -1).all()
    # one sphere, forced BVH (root is a leaf); exact tie between two coincident spheres: later wins
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        w = S.sceneobject_list()
        m = S.lambertian((0.5, 0.5, 0.5))
        w.add(S.sphere((0, 0, 0), 1, m))
        ctx.upload_scene(S.flatten(w), accel=accel)
        h = ctx.trace_primary(np.array([[0, 0, 5, 0, 0, -2], [0, 0, 0, 0, 1, 0], [3, 0, 5, 0, 0, -1]], np.float32))
        assert list(h["id"]) == [0, 0, -1]
        np.testing.assert_allclose(h["t"][:2], [2.0, 1.0], rtol=1e-6)
        assert list(h["front_face"][:2]) == [1, 0]
        w.add(S.sphere((0, 0, 0), 1, m))
        w.add(S.sphere((4, 0, 0), 1, m))
        ctx.upload_scene(S.flatten(w), accel=accel)
        assert ctx.trace_primary(np.array([[0, 0, 5, 0, 0, -1]], np.float32))["id"][0] == 1   # sphere.h:35 inclusive
    # rays along an axis (a direction component is exactly zero) through a tree with inner nodes
    w = S.sceneobject_list([S.sphere((2.5 * k, 1.0 * (k % 2), 0), 1.0, m) for k in range(12)])
    flat = S.flatten(w)
    rays = np.array([[2.5 * k, 1.0 * (k % 2), 5, 0, 0, -1] for k in range(12)] +
                    [[-5, 0, 0, 1, 0, 0], [30, 1, 0, -1, 0, 0], [5, 9, 0, 0, -1, 0], [1.25, 0.5, 5, 0, 0, -1]], np.float32)
    want = oracle.trace(flat, rays.astype(np.float64))
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE, api.RT_ACCEL_LIST):
        ctx.upload_scene(flat, accel=accel)
        h = ctx.trace_primary(rays)
        assert list(h["id"]) == list(want["id"]) == list(range(12)) + [0, 11, 2, -1]
        np.testing.assert_allclose(h["t"], want["t"], rtol=1e-5)
    # ray parallel to a rect's plane and ray hitting exactly its edge
    w = S.sceneobject_list()
    w.add(S.xz_rect(-1, 1, -1, 1, 0.0, S.lambertian((0.5, 0.5, 0.5))))
    flat = S.flatten(w)
    ctx.upload_scene(flat)
    rays = np.array([[0, 0, 5, 0, 0, -1], [-5, 1, 0, 1, 0, 0], [1, 2, 0, 0, -1, 0], [1.0001, 2, 0, 0, -1, 0]], np.float32)
    h = ctx.trace_primary(rays)
    want = oracle.trace(flat, rays.astype(np.float64))
    assert list(h["id"]) == list(want["id"]) == [-1, -1, 0, -1]
    # a far bigger sphere than wide_radius plus a negative-radius (hollow) sphere: ids and normals
    w = S.sceneobject_list()
    w.add(S.sphere((0, -5000, 0), 5000, S.lambertian((0.5, 0.5, 0.5))))
    w.add(S.sphere((0, 1, 0), 1.0, S.dielectric(1.5)))
    w.add(S.sphere((0, 1, 0), -0.9, S.dielectric(1.5)))
    flat = S.flatten(w)
    ctx.upload_scene(flat)
    rng = np.random.default_rng(3)
    rays = _f32(np.concatenate([np.tile([0, 2, 6], (500, 1)), _unit(np.tile([0, -0.2, -1], (500, 1)) + 0.2 * rng.normal(size=(500, 3)))], 1))
    _check_hits(ctx.trace_primary(rays), oracle.trace(flat, rays))
    ctx.close()


def test_invalid_scenes_are_rejected(api, S):
    ctx = api.Context(0)
    w = S.sceneobject_list()
    w.add(S.sphere((0, 0, 0), 1, S.lambertian((0.5, 0.5, 0.5))))
    flat = S.flatten(w)
    flat.prim_mat = np.array([5], np.int32)
    with pytest.raises(api.RtError) as e:
        ctx.upload_scene(flat)
    assert e.value.code == api.RT_ERR_INVALID
    flat = S.flatten(w)
    flat.prim_geom[0, 3] = 0.0
    with pytest.raises(api.RtError):
        ctx.upload_scene(flat)
    ctx.close()


# ---- the BVH stress scene itself (10^6 spheres in the bench; 2 x 10^4 here so the checker can scan it) ----

def _stress_crop(S):
    """The part of the 10^6-sphere scene of the bench that the lower half of its frame looks at: the
    same spheres, materials and textures at the scene's own density (some 2.5 x 10^4 spheres, which the
    checker's linear scan can afford; rays that leave the crop see the sky instead of the rest)."""
    import dataclasses
    full = S.stress_flat(1_000_000)
    g = full.prim_geom
    keep = (np.abs(g[:, 0]) < 70) & (g[:, 2] > -115) & (g[:, 2] < 110)
    keep_mat = np.unique(full.prim_mat[keep])
    remap = -np.ones(full.mat_type.shape[0], np.int64)
    remap[keep_mat] = np.arange(keep_mat.shape[0])
    return dataclasses.replace(full, prim_type=full.prim_type[keep], prim_geom=g[keep], prim_mat=remap[full.prim_mat[keep]].astype(np.int32),
                               mat_type=full.mat_type[keep_mat], mat_param=full.mat_param[keep_mat], mat_tex=full.mat_tex[keep_mat])


def test_stress_scene_against_the_restatement(api, S, oracle):
    """Generator and camera of the 10^6-sphere configuration against the double-precision linear
    scan: closest hits with surface coordinates of 3 000 camera rays, and a 96x54 frame at 8
    samples per pixel under the same keyed stream, on either builder's tree (bounds of
    test_textured_scene: texel and checker-cell lookups flip where float and double disagree on a
    cell edge)."""
    flat = _stress_crop(S)
    assert 15_000 < flat.n_prims < 40_000
    used = flat.tex_type[flat.mat_tex[flat.mat_tex >= 0]]
    assert (used == S.RT_TEX_IMAGE).any() and (used == S.RT_TEX_CHECKER).any()
    cam = S.stress_camera(float(np.float32(96 / 54))).desc()
    cd = oracle.camera(**cam)
    rng = np.random.default_rng(9)
    rays = _f32(np.array([oracle.get_ray(cd, rng.random(), 0.6 * rng.random(), 5, k, 0)[0] for k in range(3000)]))
    want = oracle.trace(flat, rays)
    assert (want["id"] >= 0).mean() > 0.3
    textured = flat.mat_tex[flat.prim_mat[want["id"][want["id"] >= 0]]] >= 0
    assert textured.sum() > 20
    W, H, spp = 96, 54, 8
    sums, rays_n = oracle.render(flat, cd, W, H, spp, 50, 17, threads=8)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        _check_hits(ctx.trace_primary(rays), want, uv_abs=2e-5, allow_id_mismatch=2)
        img = ctx.render_host(api.Context.params(W, H, spp, 50, 17))
        assert _rmse(img, sums, spp) <= 0.02, _rmse(img, sums, spp)
        assert abs(ctx.stats()["rays"] - rays_n) / rays_n < 0.005
        ctx.close()


# ---- kernels built for what a scene holds against the kernel that carries every feature ---------------

@pytest.mark.parametrize("which", ["spheres", "rects", "textures", "rects+textures"])
def test_specialised_kernels_render_the_general_kernels_frame(api, S, which):
    """A scene of plainly coloured spheres, a room of rects, a field of textured spheres and a
    textured room each run in a kernel without the code of the features they lack
    (RT_FLAG_ALL_FEATURES picks the kernel with all of it): same bits, same ray count, under the
    tree walk in LDS, the walk out of HBM and the scan."""
    W, H, spp = 160, 90, 8
    if which == "spheres":
        flat, cam = S.flatten(S.rtiow_scene()), S.rtiow_camera(float(np.float32(W / H))).desc()
    elif which == "rects":
        flat, cam = S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0)), S.cornell_camera(float(np.float32(W / H))).desc()
    elif which == "textures":
        flat, cam = S.stress_flat(5000), S.stress_camera(float(np.float32(W / H))).desc()
    else:
        flat, cam = S.flatten(_textured_world(S)), dict(lookfrom=(0, 2.5, 9), lookat=(0, 1, 0), vup=(0, 1, 0), vfov=35.0,
                                                        aspect=float(np.float32(W / H)), aperture=0.05, focus_dist=9.0)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        for flags in (0, api.RT_FLAG_NO_LDS_SCENE, api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES):
            for kw in ({}, {"traversal_slice": 8}):
                a = ctx.render_host(api.Context.params(W, H, spp, 50, 5, flags=flags, **kw))
                ra = ctx.stats()["rays"]
                b = ctx.render_host(api.Context.params(W, H, spp, 50, 5, flags=flags | api.RT_FLAG_ALL_FEATURES, **kw))
                assert np.array_equal(a, b) and ra == ctx.stats()["rays"], (which, accel, flags, kw)
            assert a[..., :3].max() > 0
        ctx.close()
