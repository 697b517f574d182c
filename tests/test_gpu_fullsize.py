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


def test_reference_default_frame_against_the_compiled_reference(api, rtiow_flat, reference):
    """configs[0] on the GPU: the frame main.cc renders by default (1200x800, 500 samples per pixel, depth 50,
    main.cc:90-106).  The whole frame is bit-identical under the tree walk and the scan and adds up from integer
    sample windows; every 16th pixel of every 16th row at the first 8 samples is rendered by the compiled main.cc
    itself from the same keyed stream and agrees to the bounds of the golden frames (RMSE of mean radiance 5e-4,
    under 2 % of the pixels off by more than 1e-5, ray count of those samples within 1 %)."""
    from raytrace_amd import scene as S
    W1, H1, spp = 1200, 800, 500
    cam = S.rtiow_camera(1.5).desc()
    P = api.Context.params
    frames, rays = [], []
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_LIST):
        c = api.Context(0)
        c.upload_scene(rtiow_flat, accel=accel)
        c.set_camera(**cam)
        frames.append(c.render_host(P(W1, H1, spp, 50, 1)))
        rays.append(c.stats()["rays"])
        if accel == api.RT_ACCEL_BVH:
            a = c.render_host(P(W1, H1, 200, 50, 1, flags=api.RT_FLAG_FIXED_SUMS))
            a += c.render_host(P(W1, H1, 300, 50, 1, sample_begin=200, flags=api.RT_FLAG_FIXED_SUMS))
            assert np.array_equal((a[..., :3].astype(np.float64) * 2.0 ** -32).astype(np.float32), frames[0][..., :3])
            low = c.render_host(P(W1, H1, 8, 50, 1))
        c.close()
    assert np.array_equal(frames[0], frames[1]) and rays[0] == rays[1]
    assert (frames[0][..., 3] == spp).all() and 2.5 < rays[0] / (W1 * H1 * spp) < 3.0      # main.cc: 2.73 rays per sample
    reference.load_scene(S.rtiow_records())
    reference.set_camera(**cam)
    want, _ = reference.render_keyed(W1, H1, 8, 50, 1, step=(16, 16))
    got, ref = low[::16, ::16, :3].astype(np.float64) / 8, want[::16, ::16] / 8
    assert ref.sum() > 0
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 5e-4, np.sqrt(np.mean((got - ref) ** 2))
    assert (np.abs(got - ref).max(axis=2) > 1e-5).mean() < 0.02


def test_cornell_frame_at_1000_samples_adds_up_and_reassembles(api):
    """The Cornell configuration at its own size and sample count (1920x1080, 1000 samples per
    pixel, depth 50): the frame equals the sum of two windows of its samples taken as integer sums
    (a sample-split over ranks) bit for bit, reassembles bit for bit from 8 interleaved row-tile
    shards (a tile split over ranks), and does not depend on the tile shape; the 32.32 accumulators
    and the 32-bit unit count of a tile are run at the sample count the configuration names."""
    from raytrace_amd import scene as S
    P = api.Context.params
    Wc, Hc, spp = 1920, 1080, 1000
    flat = S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    c = api.Context(0)
    c.upload_scene(flat)
    c.set_camera(**S.cornell_camera(float(np.float32(Wc / Hc))).desc())
    ref = c.render_host(P(Wc, Hc, spp, 50, 1))
    st = c.stats()
    assert (ref[..., 3] == spp).all() and np.isfinite(ref).all() and st["samples"] == Wc * Hc * spp
    assert ref[..., :3].max() > 0 and 3.0 < st["rays"] / st["samples"] < 4.0
    whole = c.render_host(P(Wc, Hc, spp, 50, 1, flags=api.RT_FLAG_FIXED_SUMS))
    assert np.array_equal((whole[..., :3].astype(np.float64) * 2.0 ** -32).astype(np.float32), ref[..., :3])
    total = c.render_host(P(Wc, Hc, 437, 50, 1, sample_begin=0, flags=api.RT_FLAG_FIXED_SUMS))
    rays = c.stats()["rays"]
    total += c.render_host(P(Wc, Hc, spp - 437, 50, 1, sample_begin=437, flags=api.RT_FLAG_FIXED_SUMS))
    rays += c.stats()["rays"]
    assert np.array_equal(total, whole) and rays == st["rays"]
    del total, whole
    out = np.zeros_like(ref)
    rays = 0
    for g in range(8):
        part = c.render_host(P(Wc, Hc, spp, 50, 1, tiles=(4, g, 8)))
        rays += c.stats()["rays"]
        out[[r for t in range(g, Hc // 4, 8) for r in range(4 * t, 4 * t + 4)]] = part
    assert np.array_equal(out, ref) and rays == st["rays"]
    assert np.array_equal(c.render_host(P(Wc, Hc, spp, 50, 1, lanes_per_pixel=32)), ref)
    c.close()


def test_million_sphere_frame_does_not_depend_on_the_builder(api):
    """The 10^6-sphere configuration at its own size and sample count (3840x2160, 64 samples per
    pixel, depth 50): the host-built and the GPU-built tree give the same bits and the same ray
    count, and a sample of primary rays hits what the device's own linear scan over all 10^6
    spheres hits.  (The scene's generator, camera and textures meet the double-precision checker
    on a crop of the scene in test_gpu_scenes.py.)"""
    from raytrace_amd import scene as S
    flat = S.stress_flat(1_000_000)
    cam = S.stress_camera(float(np.float32(3840 / 2160))).desc()
    imgs, hits, counts = [], [], []
    rng = np.random.default_rng(3)
    rays = np.concatenate([np.tile([0, 60, -120], (512, 1)), np.tile([0, -0.45, 1], (512, 1)) + 0.3 * rng.normal(size=(512, 3))], 1).astype(np.float32)
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE, api.RT_ACCEL_LIST):
        c = api.Context(0)
        c.upload_scene(flat, accel=accel)
        c.set_camera(**cam)
        if accel != api.RT_ACCEL_LIST:
            imgs.append(c.render_host(api.Context.params(3840, 2160, 64, 50, 1)))
            counts.append(c.stats()["rays"])
            assert c.stats()["bvh_depth"] <= 46 and c.stats()["samples"] == 3840 * 2160 * 64
        hits.append(c.trace_primary(rays))
        c.close()
    assert np.array_equal(imgs[0], imgs[1]) and imgs[0][..., :3].max() > 0 and counts[0] == counts[1]
    assert (imgs[0][..., 3] == 64).all() and np.isfinite(imgs[0]).all()
    for f in ("id", "t", "front_face"):
        assert np.array_equal(hits[0][f], hits[2][f]) and np.array_equal(hits[1][f], hits[2][f]), f
    assert (hits[2]["id"] >= 0).mean() > 0.5


def test_camera_path_sequence_of_240_frames_at_256_samples(api, rtiow_flat):
    """configs[4] at its own size: the 240 frames of the camera path (cuda/main.cu:220-279, :557-591) at 1920x1080, 256
    samples per pixel, through the sequence driver with two contexts in flight (no files written).  Every frame is
    rendered, in order, with the seed and camera of its number; frames picked from either side of the path's change of
    motion at frame 200 trace exactly the rays the same frames trace when rendered alone (a frame is a pure function
    of scene, camera and parameters, whichever context and stream it runs on), and the rays per frame stay in the band
    of this view.  About 11 s."""
    import torch
    from raytrace_amd import scene as S, sequence
    Wc, Hc, spp, n = 1920, 1080, 256, 240
    ctxs = []
    for _ in range(2):
        c = api.Context(0)
        c.upload_scene(rtiow_flat)
        ctxs.append(c)
    stream = torch.cuda.Stream()
    ctxs[1].set_stream(stream.cuda_stream)
    kept = {}
    picked = (0, 57, 199, 200, 239)
    done = sequence.render_sequence(ctxs[0], S.rtiow_camera(float(np.float32(Wc / Hc))), n, Wc, Hc, spp, None, max_depth=25, seed=1,
                                    overlap_ctx=ctxs[1])
    assert [d[0] for d in done] == list(range(n))
    rays = np.array([d[1] for d in done], dtype=np.float64)
    assert (rays > 2.2 * Wc * Hc * spp).all() and (rays < 3.2 * Wc * Hc * spp).all()
    # the picked frames alone: same ray count as in the sequence (a frame is a pure function of scene, camera, parameters)
    cam = S.rtiow_camera(float(np.float32(Wc / Hc)))
    fb = torch.zeros((Hc, Wc, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for frame in range(n):
        S.update_scene(cam, frame)
        if frame in picked:
            ctxs[0].set_camera(**cam.desc())
            ctxs[0].render(api.Context.params(Wc, Hc, spp, 25, 1 + frame), fb.data_ptr())
            st = ctxs[0].stats()
            assert st["rays"] == done[frame][1], frame
            kept[frame] = bool(torch.isfinite(fb).all()) and float(fb[..., 3].min()) == spp
    assert all(kept.values()) and len(kept) == len(picked)
    for c in ctxs:
        c.close()
