"""rt_trace_primary_noalloc (fixed device staging inside the library, no device allocation in the
call) against rt_trace_primary, through the C-ABI: the same hits bit for bit, on two calls in a row
of which the second is shorter, so that what the first call left in the array is shown not to
leak into the second."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CAM = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.5,
           aperture=0.1, focus_dist=10.0)


def _same(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("accel", ["bvh", "list"])
def test_noalloc_trace_equals_trace(accel, rtiow_flat):
    from raytrace_amd import api
    g = load_golden("primary_rays.npz")
    rays = np.ascontiguousarray(g["rays"], np.float32).reshape(-1, 6)
    t_min = float(g["t_min"])
    ctx = api.Context(0)
    ctx.upload_scene(rtiow_flat, accel=api.RT_ACCEL_BVH if accel == "bvh" else api.RT_ACCEL_LIST)
    want_all = ctx.trace_primary(rays, t_min)
    assert np.array_equal(want_all["id"], g["id"])               # the existing entry point is itself at the parity bar
    # first call: every fixture ray; second call: a shorter, different selection
    got_all = ctx.trace_primary(rays, t_min, noalloc=True)
    assert _same(got_all, want_all)
    sel = rays[::-1][: len(rays) // 3][::2].copy()
    want_sel = ctx.trace_primary(sel, t_min)
    got_sel = ctx.trace_primary(sel, t_min, noalloc=True)
    assert _same(got_sel, want_sel)
    assert not _same(got_sel, got_all[: len(sel)])                # it really is another answer
    # a second context on the same GPU shares the array: one call after the other is fine
    other = api.Context(0)
    other.upload_scene(rtiow_flat, accel=api.RT_ACCEL_LIST)
    assert np.array_equal(other.trace_primary(sel, t_min, noalloc=True)["id"], want_sel["id"])
    assert _same(ctx.trace_primary(rays, t_min, noalloc=True), want_all)
    other.close()
    ctx.close()


def test_noalloc_trace_limits(rtiow_flat):
    from raytrace_amd import api
    ctx = api.Context(0)
    n_max = api.TRACE_NOALLOC_MAX_RAYS
    rng = np.random.default_rng(3)
    rays = np.concatenate([np.tile(np.float32([6, 3, -13]), (n_max + 1, 1)),
                           rng.normal(size=(n_max + 1, 3)).astype(np.float32) * 0.2 + np.float32([-6, -3, 13])], axis=1)
    hits = np.zeros(n_max + 1, api.HIT_DTYPE)
    fn = api.lib().rt_trace_primary_noalloc
    assert fn(ctx._h, rays.ctypes.data, 10, ctypes.c_float(0.001), hits.ctypes.data) == api.RT_ERR_NO_SCENE
    ctx.upload_scene(rtiow_flat)
    # one ray too many is refused with a message that names the limit; nothing is written
    assert fn(ctx._h, rays.ctypes.data, n_max + 1, ctypes.c_float(0.001), hits.ctypes.data) == api.RT_ERR_UNSUPPORTED
    assert str(n_max).encode() in api.lib().rt_last_error(ctx._h)
    assert not hits["id"].any()
    assert fn(ctx._h, rays.ctypes.data, -1, ctypes.c_float(0.001), hits.ctypes.data) == api.RT_ERR_INVALID
    assert fn(ctx._h, None, 5, ctypes.c_float(0.001), hits.ctypes.data) == api.RT_ERR_INVALID
    assert fn(ctx._h, rays.ctypes.data, 0, ctypes.c_float(0.001), hits.ctypes.data) == api.RT_OK
    # exactly the maximum fills the array to its last byte; then a short call
    full = ctx.trace_primary(rays[:n_max], noalloc=True)
    assert _same(full, ctx.trace_primary(rays[:n_max]))
    assert (full["id"] >= 0).sum() > n_max // 4
    few = ctx.trace_primary(rays[n_max - 7: n_max], noalloc=True)
    assert _same(few, full[n_max - 7:])
    ctx.close()


def test_noalloc_trace_from_two_threads_with_a_context_each(rtiow_flat):
    """Two host threads, each with its own context on the same GPU (the way a sequence driver or a
    per-thread worker holds them), call the entry point at the same time: the GPU's one staging array
    is handed from call to call inside the library, so every answer equals rt_trace_primary's."""
    import threading
    from raytrace_amd import api
    g = load_golden("primary_rays.npz")
    rays = np.ascontiguousarray(g["rays"], np.float32).reshape(-1, 6)
    t_min = float(g["t_min"])
    sets = [rays, rays[::-1].copy()]                                 # different rays per thread: a mix-up would show
    ctxs, wants = [], []
    for k in range(2):
        c = api.Context(0)
        c.upload_scene(rtiow_flat, accel=api.RT_ACCEL_BVH if k == 0 else api.RT_ACCEL_LIST)
        ctxs.append(c)
        wants.append(c.trace_primary(sets[k], t_min))
    assert not _same(wants[0], wants[1])
    bad = []

    def work(k):
        for _ in range(40):
            if not _same(ctxs[k].trace_primary(sets[k], t_min, noalloc=True), wants[k]):
                bad.append(k)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad
    for c in ctxs:
        c.close()
