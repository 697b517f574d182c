"""GPU parity, second part: checks that decide the rarely taken branches per pixel (a path of at
most two rays is decidable in single precision), frames of the camera path against the checker,
a render through a texture decoded from a JPEG file, the integer sums of sample windows, and the
scenes whose ids do not fit the packed LDS records."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLD, load_golden

pytestmark = pytest.mark.gpu

CAM = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.5,
           aperture=0.1, focus_dist=10.0)


@pytest.fixture(scope="module")
def api():
    from raytrace_amd import api as A
    return A


@pytest.fixture(scope="module")
def S():
    from raytrace_amd import scene
    return scene


# ---- depth limit 2: camera ray, first hit, one scatter, second hit -----------------------------------

@pytest.mark.parametrize("accel", ["bvh", "list"])
def test_depth_limit_two_pins_every_pixel(api, S, rtiow_flat, oracle, reference, accel):
    """main.cc:10-35 with depth 2: a sample is a camera ray (camera.h:40-45, in_unit_disk), its
    closest hit, ONE scatter (material.h:21-91: lambertian's unit vector and near-zero guard,
    metal's fuzz sphere and absorb test, glass's reflect-or-refract draw) and one more closest
    hit; the result is attenuation x sky or black.  No long chain amplifies rounding, so the
    single-precision kernels decide every branch like the double-precision reference: the pixels
    of the golden view agree to 1e-5 and the ray count is equal.  The one kind of sample single
    precision cannot decide is a ray that passes a sphere's silhouette closer than its rounding
    (here: one camera ray of 153 600 samples grazes a sphere with a half chord of 4e-4 of its
    distance, 1.9e-6 of the radius inside the silhouette): at most two pixels may differ for that.
    The golden side is the compiled main.cc itself (same keyed stream), and the restatement agrees
    with it bit for bit."""
    W, H, spp, depth, seed = 120, 80, 16, 2, 5
    reference.load_scene(S.rtiow_records())
    reference.set_camera(**{k: CAM[k] for k in CAM})
    want, rays = reference.render_keyed(W, H, spp, depth, seed)
    sums, rays2 = oracle.render(rtiow_flat, oracle.camera(**CAM), W, H, spp, depth, seed, threads=8)
    assert np.array_equal(sums, want) and rays2 == rays
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat, accel=api.RT_ACCEL_BVH if accel == "bvh" else api.RT_ACCEL_LIST)
    ctx.set_camera(**CAM)
    img = ctx.render_host(api.Context.params(W, H, spp, depth, seed))
    st = ctx.stats()
    err = np.abs(img[..., :3].astype(np.float64) / spp - want / spp)
    off = np.argwhere(err.max(axis=2) > 1e-5)
    assert len(off) <= 2, (float(err.max()), off[:5])
    assert abs(st["rays"] - rays) <= 2 * len(off)          # equal, but for the samples of those pixels
    # all three materials are among the first hits, absorbed metal scatters and refractions included
    assert 1.5 * W * H * spp < rays < 2 * W * H * spp
    ctx.close()


# ---- primary hit ids through the frame kernels themselves ---------------------------------------------------

def test_primary_hit_ids_through_every_frame_kernel(api, S, oracle):
    """rt_trace_primary walks the tree in HBM in one go; a frame is rendered by other instances of
    the walk (nodes in LDS read through sign-carrying addresses, time slices, either builder's tree,
    the scan).  Here every sphere of the reference's scene emits a colour that spells its list
    position, the depth limit is 1 and there is one sample per pixel, so a pixel of the frame IS the
    id of its camera ray's closest hit: each kernel's frame is decoded and compared with the
    checker's, pixel by pixel (the camera ray included: jitter and lens draws of camera.h:40-45)."""
    rec = S.rtiow_records()
    w = S.sceneobject_list()
    for k, o in enumerate(rec):
        w.add(S.sphere(o[0:3], o[3], S.diffuse_light(((k & 31) / 32 + 1 / 64, ((k >> 5) & 31) / 32 + 1 / 64, 1.0))))
    flat = S.flatten(w, background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    W, H, seed = 240, 160, 12
    cam = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.5, aperture=0.1, focus_dist=10.0)

    def decode(img):
        hit = img[..., 2] > 0.5
        ids = np.floor(img[..., 0] * 32).astype(np.int64) + 32 * np.floor(img[..., 1] * 32).astype(np.int64)
        return np.where(hit, ids, -1)

    sums, rays = oracle.render(flat, oracle.camera(**cam), W, H, 1, 1, seed, threads=8)
    want = decode(sums)
    assert rays == W * H and len(np.unique(want)) > 100 and (want == 0).mean() > 0.3        # the ground is id 0
    P = api.Context.params
    frames = []
    for accel, kw in ((api.RT_ACCEL_BVH, {}), (api.RT_ACCEL_BVH, dict(flags=api.RT_FLAG_NO_LDS_SCENE)),
                      (api.RT_ACCEL_BVH, dict(flags=api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES)),
                      (api.RT_ACCEL_BVH_DEVICE, dict(traversal_slice=4, flags=api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES)),
                      (api.RT_ACCEL_BVH, dict(traversal_slice=3)), (api.RT_ACCEL_BVH, dict(traversal_slice=5, leaf_quorum=1, flags=api.RT_FLAG_NO_LDS_SCENE)),
                      (api.RT_ACCEL_BVH_DEVICE, {}), (api.RT_ACCEL_BVH_DEVICE, dict(traversal_slice=2, leaf_quorum=-1)), (api.RT_ACCEL_LIST, {})):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        img = ctx.render_host(P(W, H, 1, 1, seed, **kw))
        assert ctx.stats()["rays"] == W * H
        got = decode(img)
        # a camera ray that grazes a silhouette closer than single precision resolves may fall on either side
        assert (got != want).sum() <= 4, (accel, kw, int((got != want).sum()))
        frames.append(img)
        ctx.close()
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])


# ---- camera path (cuda/main.cu:220-279) against the checker ---------------------------------------------

@pytest.mark.parametrize("frame", [150, 450])
def test_camera_path_frame_matches_the_checker(api, S, rtiow_flat, oracle, frame):
    """Frames of the reference's camera animation: 150 lies in the first yaw sweep, 450 in the
    pitch-and-translate segment (cuda/main.cu:227-262).  The camera of that frame goes to the
    kernels and to the checker; image and ray count must agree as on the static view."""
    W, H, spp, depth, seed = 192, 108, 16, 50, 1000 + frame
    cam = S.rtiow_camera(W / H)
    for f in range(frame + 1):
        S.update_scene(cam, f)
    assert cam.look_from != (6.0, 3.0, -13.0)
    if frame == 450:
        assert cam.angles[0] != 0.0 and abs(cam.look_from[1] - 3.0) > 0.5       # pitched and moved
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat)
    ctx.set_camera(**cam.desc())
    img = ctx.render_host(api.Context.params(W, H, spp, depth, seed))
    st = ctx.stats()
    sums, rays = oracle.render(rtiow_flat, oracle.camera(**cam.desc()), W, H, spp, depth, seed, threads=8)
    rmse = np.sqrt(np.mean((img[..., :3].astype(np.float64) / spp - sums / spp) ** 2))
    assert rmse <= 5e-4, rmse
    assert abs(st["rays"] - rays) / rays < 2e-4, (st["rays"], rays)
    ctx.close()


def test_camera_path_frame_at_full_size(api, S, rtiow_flat, oracle):
    """One 1920x1080 frame of the path at its 256 samples per pixel: the frame does not depend on the
    closest-hit strategy or on how it is cut into shards and sample windows, and a stratified
    sample of it agrees with the checker under the same camera."""
    W, H, spp, depth, frame = 1920, 1080, 256, 50, 150
    seed = 1000 + frame
    cam = S.rtiow_camera(float(np.float32(W / H)))
    for f in range(frame + 1):
        S.update_scene(cam, f)
    P = api.Context.params
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat)
    ctx.set_camera(**cam.desc())
    ref = ctx.render_host(P(W, H, spp, depth, seed))
    st = ctx.stats()
    assert (ref[..., 3] == spp).all() and np.isfinite(ref).all() and st["samples"] == W * H * spp
    # GPU-built tree, the same bits
    other = api.Context(0)
    other.upload_scene(rtiow_flat, accel=api.RT_ACCEL_BVH_DEVICE)
    other.set_camera(**cam.desc())
    assert np.array_equal(other.render_host(P(W, H, spp, depth, seed)), ref) and other.stats()["rays"] == st["rays"]
    other.close()
    # frames + sample-splits: four sample windows as integer sums add up to the frame bit for bit
    total = np.zeros((H, W, 4), np.uint64)
    for k in range(4):
        total += ctx.render_host(P(W, H, spp // 4, depth, seed, sample_begin=k * (spp // 4), flags=api.RT_FLAG_FIXED_SUMS))
    whole = ctx.render_host(P(W, H, spp, depth, seed, flags=api.RT_FLAG_FIXED_SUMS))
    assert np.array_equal(total, whole)
    assert np.array_equal((whole[..., :3].astype(np.float64) * 2.0 ** -32).astype(np.float32), ref[..., :3])
    # stratified sample against the checker: every 16th pixel of every 16th row, 8 samples
    img = ctx.render_host(P(W, H, 8, depth, seed))
    sums, _ = oracle.render(rtiow_flat, oracle.camera(**cam.desc()), W, H, 8, depth, seed, step=(16, 16), threads=8)
    got, want = img[::16, ::16, :3].astype(np.float64) / 8, sums[::16, ::16] / 8
    assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-3
    assert (np.abs(got - want).max(axis=2) > 1e-5).mean() < 0.02
    ctx.close()


# ---- load_texture (cuda/main.cu:385-436) end to end: JPEG file -> texels -> render ------------------------------

def test_render_through_a_jpeg_file(api, S, oracle):
    """The file goes through rt_load_jpeg_file (the stand-in for stbi_load, cuda/main.cu:393); its
    texels equal the bytes the reference's vendored decoder produced for the same file; the
    scene rendered from them agrees with the checker, which samples those same texels
    (cuda/texture.h:63-76)."""
    path = os.path.join(GOLD, "texture_base420_48x40.jpg")
    z = load_golden("jpeg_cases.npz")
    assert open(path, "rb").read() == z["base420_48x40/jpeg"].tobytes()
    tex = S.load_texture(path)
    assert (tex.texture_width, tex.texture_height) == (48, 40)
    assert np.array_equal(tex.texels, z["base420_48x40/rgb"])          # stb_image's bytes
    w = S.sceneobject_list()
    w.add(S.sphere((0, -1000, 0), 1000, S.lambertian((0.5, 0.5, 0.5))))
    w.add(S.sphere((0, 1, 0), 1.0, S.lambertian(tex)))                   # the reference's earth (cuda/main.cu:101)
    w.add(S.sphere((-2.2, 1, 0), 1.0, S.diffuse_light(tex)))
    w.add(S.sphere((2.2, 1, 0), 1.0, S.metal((0.8, 0.8, 0.8), 0.0)))
    w.add(S.xy_rect(-4, 4, 0, 3, -2.5, S.lambertian_bg(tex)))            # a skybox face (cuda/main.cu:164-170)
    flat = S.flatten(w)
    cam = dict(lookfrom=(0, 2, 8), lookat=(0, 1, 0), vup=(0, 1, 0), vfov=35.0, aspect=1.5, aperture=0.0, focus_dist=8.0)
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    cd = oracle.camera(**cam)
    # depth 1: a pixel sees an emitter's texel, or nothing.  Only samples on a texel edge may differ.
    W, H = 120, 80
    one = ctx.render_host(api.Context.params(W, H, 1, 1, 7))
    sums, _ = oracle.render(flat, cd, W, H, 1, 1, 7, threads=4)
    lit = sums.max(axis=2) > 0
    assert lit.mean() > 0.25
    assert (np.abs(one[..., :3] - sums).max(axis=2) > 1e-6).mean() < 0.01
    texel_values = np.unique(tex.texels.reshape(-1, 3), axis=0).astype(np.float32) / np.float32(255)
    seen = np.unique(one[..., :3][lit].reshape(-1, 3), axis=0)
    assert all((np.abs(texel_values - s).max(axis=1) < 1e-6).any() for s in seen[:200])     # every lit pixel is a texel of the file
    # the whole path
    spp = 32
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 9))
    sums, rays = oracle.render(flat, cd, W, H, spp, 50, 9, threads=8)
    rmse = float(np.sqrt(np.mean((img[..., :3].astype(np.float64) / spp - sums / spp) ** 2)))
    assert rmse <= 0.02, rmse
    assert abs(ctx.stats()["rays"] - rays) / rays < 0.005
    ctx.close()


# ---- integer sums (RT_FLAG_FIXED_SUMS) and rt_resolve_sums ---------------------------------------------------------------

def test_fixed_sums_of_sample_windows_resolve_to_the_frame(api, rtiow_flat):
    import torch
    P = api.Context.params
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat)
    ctx.set_camera(**CAM)
    W, H, spp = 150, 90, 24
    frame = ctx.render_host(P(W, H, spp, 50, 3))
    mean = ctx.render_host(P(W, H, spp, 50, 3, flags=api.RT_FLAG_MEAN))
    # windows of unequal length, in any order, as a sample-split over ranks would add them
    windows = [(16, 8), (0, 5), (5, 11)]
    total = torch.zeros((H, W, 4), dtype=torch.int64, device="cuda")
    part = torch.zeros((H, W, 4), dtype=torch.int64, device="cuda")
    for begin, n in windows:
        torch.cuda.synchronize()
        ctx.render(P(W, H, n, 50, 3, sample_begin=begin, flags=api.RT_FLAG_FIXED_SUMS), part.data_ptr())
        ctx.stats()                                              # waits for the context's stream
        total += part
    assert (total[..., 3] == spp).all()
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.resolve_sums(total.data_ptr(), W * H, out.data_ptr())
    ctx.stats()
    assert np.array_equal(out.cpu().numpy(), frame)
    ctx.resolve_sums(total.data_ptr(), W * H, out.data_ptr(), mean=True)
    ctx.stats()
    assert np.array_equal(out.cpu().numpy(), mean)
    # shards of the integer image reassemble like the float one
    whole = ctx.render_host(P(W, H, spp, 50, 3, flags=api.RT_FLAG_FIXED_SUMS))
    assert np.array_equal(whole, total.cpu().numpy().view(np.uint64))
    re = np.zeros_like(whole)
    for g in range(3):
        rows = [r for t in range(g, (H + 3) // 4, 3) for r in range(4 * t, min(4 * t + 4, H))]
        re[rows] = ctx.render_host(P(W, H, spp, 50, 3, tiles=(4, g, 3), flags=api.RT_FLAG_FIXED_SUMS))
    assert np.array_equal(re, whole)
    with pytest.raises(api.RtError) as e:
        ctx.render_host(P(W, H, spp, 50, 3, flags=api.RT_FLAG_FIXED_SUMS | api.RT_FLAG_MEAN))
    assert e.value.code == api.RT_ERR_INVALID
    ctx.close()


# ---- rt_render_params layouts and traversal tuning fields ------------------------------------------------------------------

def test_params_layouts_and_tuning_fields(api, rtiow_flat):
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat)
    ctx.set_camera(**CAM)
    P = api.Context.params
    W, H, spp = 96, 64, 8
    ref = ctx.render_host(P(W, H, spp, 50, 2))
    # the layout that ends before the tuning fields is accepted and means "defaults"
    p = P(W, H, spp, 50, 2, traversal_slice=3, leaf_quorum=7)      # ignored: not part of the short layout
    p.struct_size = api.RenderParams.traversal_slice.offset
    out = np.zeros((H, W, 4), np.float32)
    assert api.lib().rt_render_host(ctx._h, ctypes.byref(p), out.ctypes.data) == api.RT_OK
    assert np.array_equal(out, ref)
    p.struct_size = api.RenderParams.stack_lds_entries.offset          # the layout with two tuning fields
    assert api.lib().rt_render_host(ctx._h, ctypes.byref(p), out.ctypes.data) == api.RT_OK
    assert np.array_equal(out, ref)
    p.struct_size = 40
    assert api.lib().rt_render_host(ctx._h, ctypes.byref(p), out.ctypes.data) == api.RT_ERR_INVALID
    # any slice and quorum: the same frame, the same rays
    rays = ctx.stats()["rays"]
    for sl, q in ((-1, 0), (1, 0), (4, -1), (7, 1), (64, 5)):
        assert np.array_equal(ctx.render_host(P(W, H, spp, 50, 2, traversal_slice=sl, leaf_quorum=q)), ref), (sl, q)
        assert ctx.stats()["rays"] == rays
    with pytest.raises(api.RtError):
        ctx.render_host(P(W, H, spp, 50, 2, traversal_slice=-2))
    # out of range is an error, not a silent clamp: a slice beyond 65534 steps, a quorum beyond a wavefront's 64 lanes
    for kw in (dict(traversal_slice=65535), dict(leaf_quorum=65), dict(leaf_quorum=2 ** 31 - 1), dict(leaf_quorum=-2)):
        with pytest.raises(api.RtError) as e:
            ctx.render_host(P(W, H, spp, 50, 2, **kw))
        assert e.value.code == api.RT_ERR_INVALID, kw
    assert np.array_equal(ctx.render_host(P(W, H, spp, 50, 2, traversal_slice=65534, leaf_quorum=64)), ref)
    ctx.close()


def test_stats_are_written_up_to_the_callers_struct_size(api, rtiow_flat):
    """rt_get_stats writes no more than the caller's struct_size (so rt_stats may grow at its end), and
    refuses a struct without one."""
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat)
    ctx.set_camera(**CAM)
    ctx.render_host(api.Context.params(64, 48, 4, 50, 2))
    full = ctx.stats()
    assert full["rays"] > 64 * 48 * 4 and full["struct_size"] == ctypes.sizeof(api.Stats)
    first = api.Stats.bvh_coarse_nodes.offset + 4                   # the struct as first published with a size
    buf = (ctypes.c_uint8 * (ctypes.sizeof(api.Stats) + 64))()
    ctypes.memset(buf, 0xAB, len(buf))
    st = api.Stats.from_buffer(buf)
    st.struct_size = first
    assert api.lib().rt_get_stats(ctx._h, ctypes.byref(st)) == api.RT_OK
    assert st.rays == full["rays"] and st.bvh_depth == full["bvh_depth"] and st.struct_size == first
    assert all(b == 0xAB for b in bytes(buf)[first:])               # nothing behind the caller's struct was touched
    for bad in (0, 8, first - 4):
        st.struct_size = bad
        assert api.lib().rt_get_stats(ctx._h, ctypes.byref(st)) == api.RT_ERR_INVALID
    st.struct_size = len(buf)                                       # a newer caller: only what this library knows is written
    assert api.lib().rt_get_stats(ctx._h, ctypes.byref(st)) == api.RT_OK and st.struct_size == ctypes.sizeof(api.Stats)
    assert all(b == 0xAB for b in bytes(buf)[ctypes.sizeof(api.Stats):])
    ctx.close()


# ---- range of a pixel's sum (32.32 fixed point against main.cc:121's double) --------------------------------------------------

def test_pixel_sums_are_refused_before_they_can_wrap(api, S):
    """An emitter of 10^7 in front of the camera: 256 samples sum to 2.56e9 < 2^32 and the pixel holds
    exactly that; 512 samples could reach 5.12e9 >= 2^32 and are refused with RT_ERR_UNSUPPORTED
    (the reference's double sum does not wrap; a silent wrap would)."""
    w = S.sceneobject_list()
    w.add(S.xy_rect(-2, 2, -2, 2, 0.0, S.diffuse_light((1.0e7, 0.5e7, 10.0))))
    flat = S.flatten(w, background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
    cam = dict(lookfrom=(0, 0, 3), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.0, aperture=0.0, focus_dist=3.0)
    ctx = api.Context(0)
    ctx.upload_scene(flat)
    ctx.set_camera(**cam)
    P = api.Context.params
    img = ctx.render_host(P(16, 16, 256, 5, 1))
    assert (img[..., 0] == np.float32(256 * 1.0e7)).all() and (img[..., 1] == np.float32(256 * 0.5e7)).all() and (img[..., 2] == 2560.0).all()
    assert (ctx.render_host(P(16, 16, 429, 5, 1))[..., 0] == np.float32(429 * 1.0e7)).all()       # 4.29e9 < 2^32
    for kw in (dict(spp=512), dict(spp=430), dict(spp=30, sample_begin=400)):
        with pytest.raises(api.RtError) as e:
            ctx.render_host(P(16, 16, kw["spp"], 5, 1, sample_begin=kw.get("sample_begin", 0)))
        assert e.value.code == api.RT_ERR_UNSUPPORTED and "2^32" in str(e.value), kw
    # the same test sees a bright solid background and a bright texture
    dim = S.sceneobject_list()
    dim.add(S.sphere((0, 0, 0), 0.5, S.lambertian((0.5, 0.5, 0.5))))
    ctx.upload_scene(S.flatten(dim, background=S.RT_BG_SOLID, bg_color=(3.0e7, 0, 0)))
    with pytest.raises(api.RtError):
        ctx.render_host(P(16, 16, 512, 5, 1))
    assert np.isfinite(ctx.render_host(P(16, 16, 128, 5, 1))).all()
    w3 = S.sceneobject_list()
    w3.add(S.xy_rect(-2, 2, -2, 2, 0.0, S.diffuse_light(S.checker_texture((1.0, 1.0, 1.0), (2.0e7, 0.0, 0.0)))))
    ctx.upload_scene(S.flatten(w3, background=S.RT_BG_SOLID, bg_color=(0, 0, 0)))
    with pytest.raises(api.RtError):
        ctx.render_host(P(16, 16, 256, 5, 1))
    assert np.isfinite(ctx.render_host(P(16, 16, 128, 5, 1))).all()
    ctx.close()


# ---- list positions and material indices beyond 16 bits ---------------------------------------------------------------------

def test_ids_beyond_16_bits_keep_their_materials(api, S, oracle):
    """A few balls at the END of a list of 70 000 rects, every primitive with a material of its own:
    list positions and material indices exceed 65535, which the packed LDS copy of a short sphere
    list cannot hold; such a scene must be read from the full-width records."""
    n_rects = 70000
    balls = [((0.0, 1.0, 0.0), 1.0), ((2.3, 1.0, 0.2), 1.0), ((-2.3, 1.0, -0.2), 1.0), ((0.5, 0.4, 2.0), 0.4)]
    n = n_rects + len(balls)
    ptype = np.full(n, S.RT_XZ_RECT, np.int32)
    ptype[n_rects:] = S.RT_SPHERE
    geom = np.zeros((n, 6))
    # rects: tiny tiles far below the view, except the first, which is the floor
    k = np.arange(n_rects)
    geom[:n_rects, 0] = -50 + (k % 300) * 0.01
    geom[:n_rects, 1] = geom[:n_rects, 0] + 0.005
    geom[:n_rects, 2] = -50 + (k // 300) * 0.01
    geom[:n_rects, 3] = geom[:n_rects, 2] + 0.005
    geom[:n_rects, 4] = -30.0
    geom[0, :5] = (-20, 20, -20, 20, 0.0)
    for i, (c, r) in enumerate(balls):
        geom[n_rects + i, :4] = (*c, r)
    mat_type = np.full(n, S.RT_LAMBERTIAN, np.int32)
    param = np.zeros((n, 5))
    param[:, :3] = 0.5
    colours = [(0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9)]
    for i, c in enumerate(colours):
        param[n_rects + i, :3] = c
    mat_type[n - 1] = S.RT_DIFFUSE_LIGHT
    param[n - 1, :3] = (2.0, 1.5, 1.0)
    flat = S.FlatScene(prim_type=ptype, prim_geom=geom, prim_mat=np.arange(n, dtype=np.int32), mat_type=mat_type, mat_param=param,
                       mat_tex=-np.ones(n, np.int32), tex_type=np.zeros(0, np.int32), tex_color=np.zeros((0, 3)),
                       tex_a=np.zeros(0, np.int32), tex_b=np.zeros(0, np.int32), tex_off=np.zeros(0, np.int64),
                       texels=np.zeros(0, np.uint8))
    # only the floor and the balls matter to the picture: the checker gets those (same list order)
    keep = np.concatenate([[0], np.arange(n_rects, n)])
    small = S.FlatScene(prim_type=ptype[keep], prim_geom=geom[keep], prim_mat=np.arange(len(keep), dtype=np.int32),
                        mat_type=mat_type[keep], mat_param=param[keep], mat_tex=-np.ones(len(keep), np.int32),
                        tex_type=flat.tex_type, tex_color=flat.tex_color, tex_a=flat.tex_a, tex_b=flat.tex_b,
                        tex_off=flat.tex_off, texels=flat.texels)
    cam = dict(lookfrom=(0, 2.5, 9), lookat=(0, 0.8, 0), vup=(0, 1, 0), vfov=35.0, aspect=1.5, aperture=0.0, focus_dist=9.0)
    W, H, spp = 48, 32, 4
    sums, rays = oracle.render(small, oracle.camera(**cam), W, H, spp, 4, 6, threads=8)
    imgs = []
    for accel in (api.RT_ACCEL_LIST, api.RT_ACCEL_BVH):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        hits = ctx.trace_primary(np.array([[0, 2.5, 9, 0, -1.5, -9], [0, 2.5, 9, 2.3, -1.5, -8.8], [0, 2.5, 9, 0.5, -2.1, -7]], np.float32))
        assert list(hits["id"]) == [n_rects, n_rects + 1, n_rects + 3]
        for flags in (0, api.RT_FLAG_NO_LDS_SCENE):
            imgs.append(ctx.render_host(api.Context.params(W, H, spp, 4, 6, flags=flags)))
            assert ctx.stats()["rays"] == rays
        ctx.close()
    for im in imgs[1:]:
        assert np.array_equal(im, imgs[0])
    err = np.abs(imgs[0][..., :3].astype(np.float64) / spp - sums / spp)
    assert (err.max(axis=2) > 1e-5).mean() < 0.02, float(err.max())
    assert imgs[0][..., 0].max() > 1.0 * spp * 0.5                 # the light's colour is seen


# ---- RT_ACCEL_AUTO falls back to the host builder when the Morton tree is too deep ------------------------------------------------

def test_auto_accel_survives_a_scene_too_deep_for_the_gpu_builder(api, S):
    """65 600 spheres of which 63 sit at 2^-k along each axis: their Morton keys differ in one
    leading bit each, a chain as deep as the key is long, deeper than the traversal stack.  Asked
    for by name the GPU builder says so; RT_ACCEL_AUTO, which picks that builder for scenes of this
    size on its own, must come back with the host-built tree instead."""
    rng = np.random.default_rng(2)
    pts = []
    for k in range(21):
        for axis in range(3):
            p = [0.0, 0.0, 0.0]
            p[axis] = 1000.0 * 2.0 ** -k
            pts.append(p)
    pts = np.array(pts)
    cloud = 990.0 + 10.0 * rng.random((65600 - len(pts), 3))
    c = np.concatenate([pts, cloud])
    n = len(c)
    r = np.concatenate([np.maximum(1e-5, 0.2 * np.linalg.norm(pts, axis=1)), np.full(n - len(pts), 0.02)])
    flat = S.FlatScene(prim_type=np.zeros(n, np.int32), prim_geom=np.concatenate([c, r[:, None], np.zeros((n, 2))], 1),
                       prim_mat=np.zeros(n, np.int32), mat_type=np.zeros(1, np.int32), mat_param=np.array([[.5, .5, .5, 0, 0]]),
                       mat_tex=-np.ones(1, np.int32), tex_type=np.zeros(0, np.int32), tex_color=np.zeros((0, 3)),
                       tex_a=np.zeros(0, np.int32), tex_b=np.zeros(0, np.int32), tex_off=np.zeros(0, np.int64), texels=np.zeros(0, np.uint8))
    rays = np.concatenate([np.tile([500.0, 400.0, -900.0], (600, 1)),
                           c[rng.integers(0, n, 600)] - [500.0, 400.0, -900.0] + 0.01 * rng.normal(size=(600, 3))], 1).astype(np.float32)
    rays[:63, 3:] = pts - [500.0, 400.0, -900.0]
    ctx = api.Context(0)
    with pytest.raises(api.RtError) as e:
        ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH_DEVICE)
    assert e.value.code == api.RT_ERR_UNSUPPORTED and b"deeper" in api.lib().rt_last_error(ctx._h)
    ctx.upload_scene(flat, accel=api.RT_ACCEL_AUTO)
    auto = ctx.trace_primary(rays)
    ctx.set_camera((500, 400, -900), (995, 995, 995), (0, 1, 0), 30.0, 1.5, 0.0, 10.0)
    ctx.render_host(api.Context.params(48, 32, 2, 5, 1))
    st = ctx.stats()
    assert st["accel"] == api.RT_ACCEL_BVH and st["bvh_depth"] <= 46
    ctx.upload_scene(flat, accel=api.RT_ACCEL_LIST)
    scan = ctx.trace_primary(rays)
    for f in ("id", "t", "front_face"):
        assert np.array_equal(auto[f], scan[f]), f
    assert (scan["id"] >= 0).mean() > 0.5 and (scan["id"][:63] >= 0).sum() > 40
    ctx.close()


# ---- 32-byte quantised nodes against the 64-byte float nodes and the scan ------------------------------------------------------

def mixed_scale_scene(S, offset, r_min=2e-4):
    """40 000 emitters with radii from r_min to 30 in a box 2 000 wide, optionally far from the
    origin, each emitting its list position; every tenth is of the smallest and sits in front of the
    camera at a distance where they cover pixels.  Returns the scene, the camera and the ids of
    those small ones."""
    rng = np.random.default_rng(11)
    n = 40000
    c = rng.random((n, 3)) * [2000.0, 40.0, 2000.0] + offset
    r = r_min * (30.0 / r_min) ** rng.random(n) ** 3
    near = np.arange(0, n, 10)
    cam_from = np.array([1000.0, 20.0, -260.0]) + offset
    s = r_min / 2e-4
    c[near] = cam_from + s * np.stack([rng.uniform(-0.02, 0.02, len(near)), rng.uniform(-0.012, 0.012, len(near)), rng.uniform(0.1, 0.2, len(near))], 1)
    r[near] = rng.uniform(r_min, 3 * r_min, len(near))
    k = np.arange(n)
    param = np.zeros((n, 5))
    param[:, 0] = (k % 256 + 0.5) / 256
    param[:, 1] = (k // 256 + 0.5) / 256
    param[:, 2] = 1.0
    flat = S.FlatScene(prim_type=np.zeros(n, np.int32), prim_geom=np.concatenate([c, r[:, None], np.zeros((n, 2))], 1),
                       prim_mat=np.arange(n, dtype=np.int32), mat_type=np.full(n, S.RT_DIFFUSE_LIGHT, np.int32), mat_param=param,
                       mat_tex=-np.ones(n, np.int32), tex_type=np.zeros(0, np.int32), tex_color=np.zeros((0, 3)),
                       tex_a=np.zeros(0, np.int32), tex_b=np.zeros(0, np.int32), tex_off=np.zeros(0, np.int64), texels=np.zeros(0, np.uint8),
                       background=S.RT_BG_SOLID, bg_color=(0.0, 0.0, 0.0))
    cam = dict(lookfrom=tuple(cam_from), lookat=tuple(np.array([1000.0, 20.0, 1000.0]) + offset), vup=(0, 1, 0), vfov=12.0, aspect=1.5,
               aperture=0.0, focus_dist=10.0)
    return flat, cam, near


@pytest.mark.parametrize("offset,r_min", [(0.0, 0.15), (3.0e4, 0.15), (0.0, 2e-4)])
def test_quantised_nodes_lose_no_hit_on_a_scene_of_mixed_scales(api, S, offset, r_min):
    """A tree read from HBM is walked through nodes whose planes are 16-bit steps of the scene's box,
    0.03 wide in this scene: a fifth of the smallest radius in the first two cases, a hundred times
    the smallest radius in the third, where the tree must therefore keep its float form.  At depth
    1 a pixel is the id of the closest hit: the walk through quantised nodes, the walk through
    float nodes and the scan of the list must agree on every pixel and every ray count, for both
    builders."""
    flat, cam, near = mixed_scale_scene(S, offset, r_min)
    W, H = 192, 128
    P = api.Context.params
    frames, rays = {}, {}
    for name, accel, flags in (("list", api.RT_ACCEL_LIST, 0),
                               ("host quantised", api.RT_ACCEL_BVH, api.RT_FLAG_NO_LDS_SCENE),
                               ("host float", api.RT_ACCEL_BVH, api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES),
                               ("device quantised", api.RT_ACCEL_BVH_DEVICE, api.RT_FLAG_NO_LDS_SCENE),
                               ("device float", api.RT_ACCEL_BVH_DEVICE, api.RT_FLAG_NO_LDS_SCENE | api.RT_FLAG_FLOAT_NODES)):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        frames[name] = ctx.render_host(P(W, H, 2, 1, 3, flags=flags | api.RT_FLAG_COUNTERS))
        rays[name] = ctx.stats()["rays"]
        if name == "host quantised":
            box_q = ctx.stats()["box_tests"]
            coarse = ctx.stats()["bvh_coarse_nodes"] / ctx.stats()["bvh_nodes"]
        if name == "host float":
            box_f = ctx.stats()["box_tests"]
        ctx.close()
    ids = np.floor(frames["list"][..., 0] / 2 * 256).astype(int) + 256 * np.floor(frames["list"][..., 1] / 2 * 256).astype(int)
    seen = np.unique(ids[frames["list"][..., 2] > 1.5])
    assert len(seen) > 300 and np.isin(seen, near).sum() > 20          # far and near, large and tiny spheres are in the picture
    for name, f in frames.items():
        assert np.array_equal(f, frames["list"]), name
        assert rays[name] == rays["list"], name
    if r_min > 0.1:
        # the quantised boxes are a little larger: a few more visits, never fewer
        assert coarse == 0 and box_f < box_q <= 1.25 * box_f, (coarse, box_f, box_q)
    else:
        # nodes over spheres far smaller than a step of the grid: the flag changes nothing
        assert 0.05 < coarse < 0.5 and box_q == box_f, (coarse, box_f, box_q)


# ---- a traversal stack that does not fit in LDS -------------------------------------------------------------------------------

def test_a_stack_that_spills_out_of_lds_loses_no_node(api, S):
    """The time-sliced walk of a tree in HBM keeps at most stack_lds_entries entries of a lane's stack
    in LDS (24 by default: what leaves six workgroups per CU) and the rest in device memory, which a
    walk reaches only in trees deeper than 22 levels.  With 2, 3 and 5 entries in LDS nearly every push
    of this 15-level tree goes to device memory: the frame and the ray count must not change, whichever
    builder made the tree, at full depth and with id-coded emitters (depth 1: a pixel is its closest hit)."""
    flat = S.stress_flat(20_000)
    cam = S.stress_camera(1.5).desc()
    W, H = 240, 160
    P = api.Context.params
    for accel in (api.RT_ACCEL_BVH, api.RT_ACCEL_BVH_DEVICE):
        ctx = api.Context(0)
        ctx.upload_scene(flat, accel=accel)
        ctx.set_camera(**cam)
        assert ctx.stats()["bvh_depth"] >= 13 and ctx.stats()["bvh_coarse_nodes"] == 0
        for depth, spp in ((50, 4), (1, 2)):
            ref = ctx.render_host(P(W, H, spp, depth, 5))
            rays = ctx.stats()["rays"]
            for entries, sl in ((2, 40), (3, 7), (5, 128), (48, 16)):      # (a 15-level tree is not time-sliced by default: ask for it)
                got = ctx.render_host(P(W, H, spp, depth, 5, stack_lds_entries=entries, traversal_slice=sl))
                assert np.array_equal(got, ref), (accel, depth, entries)
                assert ctx.stats()["rays"] == rays
        with pytest.raises(api.RtError):
            ctx.render_host(P(W, H, 2, 1, 5, stack_lds_entries=1))
        ctx.close()


def test_a_walk_left_between_rounds_of_leaves_is_the_same_walk(api, S):
    """A scene of spheres only under a tree of at most 16 levels: by default a wave leaves the walk
    between two rounds of leaves once a quarter of the lanes that came in are unfinished, and those
    lanes resume on the next trip.  The frame and its ray count are those of the walk that is never
    left (traversal_slice -1), at any share (128: left after every round), with the tree in LDS or
    read from HBM, at full depth and with the depth limit at 1."""
    flat = S.flatten(S.rtiow_scene())
    ctx = api.Context(0)
    ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH)
    ctx.set_camera(**S.rtiow_camera(1.5).desc())
    W, H = 240, 160
    P = api.Context.params
    for lds in (0, api.RT_FLAG_NO_LDS_SCENE):
        for depth, spp in ((50, 8), (1, 3)):
            whole = ctx.render_host(P(W, H, spp, depth, 3, flags=lds, traversal_slice=-1))
            rays = ctx.stats()["rays"]
            for share in (0, 1, 16, 127, 128, -1):
                got = ctx.render_host(P(W, H, spp, depth, 3, flags=lds, traversal_exit_share=share))
                assert np.array_equal(got, whole), (lds, depth, share)
                assert ctx.stats()["rays"] == rays
    ctx.close()
