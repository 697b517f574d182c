"""Size-independent properties at the full size of the headline configuration (1920x1080, 100
samples per pixel, depth 50), where the CPU checkers would take hours: the frame is a pure
function of (scene, camera, parameters), whatever the closest-hit strategy, the sharding, the
sample windows or the tile shape, and the stratified sample of it that the checker can afford
agrees with the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, SPP, DEPTH, SEED = 1920, 1080, 100, 50, 1


@pytest.fixture(scope="module")
def api():
    from raytrace_amd import api as A
    return A


@pytest.fixture(scope="module")
def frames(api, rtiow_flat):
    from raytrace_amd import scene as S
    out = {}
    for name, accel in (("bvh", api.RT_ACCEL_BVH), ("gpu_bvh", api.RT_ACCEL_BVH_DEVICE), ("list", api.RT_ACCEL_LIST)):
        c = api.Context(0)
        c.upload_scene(rtiow_flat, accel=accel)
        c.set_camera(**S.rtiow_camera(float(np.float32(W / H))).desc())
        out[name] = (c, c.render_host(api.Context.params(W, H, SPP, DEPTH, SEED)), c.stats())
    yield out
    for c, _, _ in out.values():
        c.close()


def test_every_strategy_renders_the_same_frame(frames):
    _, ref, st = frames["bvh"]
    assert (ref[..., 3] == SPP).all() and np.isfinite(ref).all() and st["samples"] == W * H * SPP
    for name in ("gpu_bvh", "list"):
        _, img, st2 = frames[name]
        assert np.array_equal(img, ref), name
        assert st2["rays"] == st["rays"]
    assert 2.5 < st["rays"] / st["samples"] < 3.0          # main.cc traces 2.72 rays per sample on this view


def test_shards_windows_and_tile_shapes_reassemble(api, frames):
    c, ref, st = frames["bvh"]
    P = api.Context.params
    out = np.zeros_like(ref)
    rays = 0
    for g in range(8):
        part = c.render_host(P(W, H, SPP, DEPTH, SEED, tiles=(4, g, 8)))
        rays += c.stats()["rays"]
        rows = [r for t in range(g, H // 4, 8) for r in range(4 * t, 4 * t + 4)]
        assert part.shape[0] == len(rows)
        out[rows] = part
    assert np.array_equal(out, ref) and rays == st["rays"]
    a = c.render_host(P(W, H, 60, DEPTH, SEED, sample_begin=0))
    b = c.render_host(P(W, H, 40, DEPTH, SEED, sample_begin=60))
    np.testing.assert_allclose(a[..., :3] + b[..., :3], ref[..., :3], rtol=2e-6, atol=1e-5)
    assert np.array_equal(c.render_host(P(W, H, SPP, DEPTH, SEED, lanes_per_pixel=16)), ref)


def test_stratified_sample_of_the_frame_agrees_with_the_checker(frames, oracle, rtiow_flat):
    """Every 16th pixel of every 16th row at the first 8 samples per pixel: the same keyed
    paths through the double-precision restatement (itself bit-exact with the compiled main.cc
    on the golden frames)."""
    from raytrace_amd import api as A, scene as S
    c, _, _ = frames["bvh"]
    img = c.render_host(A.Context.params(W, H, 8, DEPTH, SEED))
    cam = oracle.camera(**S.rtiow_camera(float(np.float32(W / H))).desc())
    sums, _ = oracle.render(rtiow_flat, cam, W, H, 8, DEPTH, SEED, step=(16, 16), threads=8)
    got, want = img[::16, ::16, :3].astype(np.float64) / 8, sums[::16, ::16] / 8
    assert want.sum() > 0
    rmse = np.sqrt(np.mean((got - want) ** 2))
    assert rmse <= 1e-3, rmse
    assert (np.abs(got - want).max(axis=2) > 1e-5).mean() < 0.02


def test_million_sphere_frame_does_not_depend_on_the_builder(api):
    """The 10^6-sphere configuration at its full 3840x2160 size (2 samples per pixel here): the
    host-built and the GPU-built tree give the same bits, and a sample of primary rays hits what
    the device's own linear scan over all 10^6 spheres hits."""
    from raytrace_amd import scene as S
    flat = S.stress_flat(1_000_000)
    cam = S.stress_camera(float(np.float32(3840 / 2160))).desc()
    imgs, hits = [], []
    rng = np.random.default_rng(3)
    rays = np.concatenate([np.tile([0, 60, -120], (512, 1)), np.tile([0, -0.45, 1], (512, 1)) + 0.3 * rng.normal(size=(512, 3))], 1).astype(np.float32)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE, api.RT_ACCEL_LIST):
        c = api.Context(0)
        c.upload_scene(flat, accel=accel)
        c.set_camera(**cam)
        if accel != api.RT_ACCEL_LIST:
            imgs.append(c.render_host(api.Context.params(3840, 2160, 2, 50, 1)))
            assert c.stats()["bvh_depth"] <= 46
        hits.append(c.trace_primary(rays))
        c.close()
    assert np.array_equal(imgs[0], imgs[1]) and imgs[0][..., :3].max() > 0
    for f in ("id", "t", "front_face"):
        assert np.array_equal(hits[0][f], hits[2][f]) and np.array_equal(hits[1][f], hits[2][f]), f
    assert (hits[2]["id"] >= 0).mean() > 0.5
