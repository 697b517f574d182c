"""Life cycle of contexts and scenes through the C-ABI: what a long-running caller (a sequence
driver, a server) does over and over must neither leak device memory nor depend on what ran before.
cuda/main.cu builds its world once and exits (main.cu:439-651), so the reference has no
counterpart of these; the expected frames are the library's own from a fresh context."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAM = dict(lookfrom=(6, 3, -13), lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=1.5,
           aperture=0.1, focus_dist=10.0)


@pytest.fixture(scope="module")
def api():
    from raytrace_amd import api as A
    return A


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_repeated_uploads_and_contexts_do_not_leak(api, rtiow_flat):
    """Scenes of either kind uploaded over one another (host tree in LDS, GPU-built tree with its
    quantised copy, the scan), contexts created and destroyed: free device memory ends where it
    started, and the frame after all of it equals the first one."""
    from raytrace_amd import scene as S
    big = S.stress_flat(70_000)
    p = api.Context.params(96, 64, 4, 50, 7)
    c = api.Context(0)
    c.upload_scene(rtiow_flat)
    c.set_camera(**CAM)
    first = c.render_host(p)
    c.trace_primary(np.tile(np.array([[6, 3, -13, -6, -3, 13]], np.float32), (1000, 1)), 0.001)   # grows the trace buffer
    base = None
    for k in range(15):
        if k == 3:
            base = _free_bytes()                            # (the runtime's own pools have grown to their size by now)
        c.upload_scene(big if k % 2 == 0 else rtiow_flat, accel=(api.RT_ACCEL_AUTO, api.RT_ACCEL_BVH, api.RT_ACCEL_LIST)[k % 3])
        c.render_host(p)
        d = api.Context(0)
        d.upload_scene(big, accel=api.RT_ACCEL_BVH_DEVICE)
        d.set_camera(**CAM)
        d.render_host(p)
        d.close()
    c.upload_scene(rtiow_flat)
    assert np.array_equal(c.render_host(p), first)
    # the allocator hands memory back in 2 MiB pieces: allow a few of them, not a scene (the big one takes 9 MB and more)
    assert abs(_free_bytes() - base) <= 8 << 20, (_free_bytes(), base)
    c.close()


def test_rejected_calls_leave_the_context_usable(api, rtiow_flat):
    """A scene that is refused (and frees what was there: rt.h says so) and parameters that are
    refused do not poison the context: the next valid upload renders the frame a fresh context renders."""
    from raytrace_amd import scene as S
    p = api.Context.params(80, 60, 3, 50, 11)
    fresh = api.Context(0)
    fresh.upload_scene(rtiow_flat)
    fresh.set_camera(**CAM)
    want = fresh.render_host(p)
    fresh.close()
    c = api.Context(0)
    c.upload_scene(rtiow_flat)
    c.set_camera(**CAM)
    bad = S.flatten(S.rtiow_scene())
    bad.prim_mat = bad.prim_mat.copy()
    bad.prim_mat[3] = 10 ** 6                               # refers outside the material array
    with pytest.raises(api.RtError):
        c.upload_scene(bad)
    for kw in (dict(lanes_per_pixel=5), dict(tiles=(0, 3, 2)), dict(flags=api.RT_FLAG_FIXED_SUMS | api.RT_FLAG_MEAN),
               dict(traversal_exit_share=129), dict(traversal_exit_share=-2), dict(stack_lds_entries=49)):
        with pytest.raises(api.RtError):
            c.render_host(api.Context.params(80, 60, 3, 50, 11, **kw))
    c.upload_scene(rtiow_flat)
    assert np.array_equal(c.render_host(p), want)
    c.close()


def test_contexts_on_threads_render_the_same_bits(api, rtiow_flat):
    """A context belongs to one thread at a time; four threads with a context each, rendering at
    once (launches from different host threads interleave on the device), get the bits one gets alone."""
    p = api.Context.params(160, 90, 8, 50, 3)
    c0 = api.Context(0)
    c0.upload_scene(rtiow_flat)
    c0.set_camera(**CAM)
    want = c0.render_host(p)
    c0.close()
    out, err = [None] * 4, []

    def work(i):
        try:
            c = api.Context(0)
            c.upload_scene(rtiow_flat, accel=api.RT_ACCEL_BVH if i % 2 else api.RT_ACCEL_LIST)
            c.set_camera(**CAM)
            for _ in range(5):
                out[i] = c.render_host(p)
            c.close()
        except Exception as e:                              # noqa: BLE001 (reported below)
            err.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not err, err
    for o in out:
        assert o is not None and np.array_equal(o, want)


def test_a_device_buffer_that_is_too_small_is_an_error_not_a_fault(api, rtiow_flat):
    """rt_render / rt_quantize / rt_resolve_sums write into bare device pointers; where the runtime
    knows the allocation, a buffer that ends too early is refused on the host (a kernel writing past
    it would fault the GPU).  Checked on raw hipMalloc'ed buffers, the case the runtime can place exactly."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    W, H = 64, 48

    def dev(nbytes):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0
        return p

    full, short = dev(W * H * 16), dev(W * H * 16 - 4096)
    # the runtime must place these pointers exactly, or the library takes them on trust (and this
    # test must not hand it a buffer that is too small)
    hip.hipMemGetAddressRange.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    base, size = ctypes.c_void_p(), ctypes.c_size_t()
    if hip.hipMemGetAddressRange(ctypes.byref(base), ctypes.byref(size), short) != 0 or size.value != W * H * 16 - 4096 \
            or base.value != short.value:
        pytest.skip("the runtime does not report the extent of a hipMalloc allocation exactly")
    c = api.Context(0)
    c.upload_scene(rtiow_flat)
    c.set_camera(**CAM)
    sums = dev(W * H * 32)
    rgb = dev(W * H * 3)
    p = api.Context.params(W, H, 2, 50, 1)
    c.render(p, full.value)                                                     # fits
    c.render(api.Context.params(W, H, 2, 50, 1, tiles=(4, 0, 2)), short.value)   # half the rows fit the shorter buffer
    for bad in (lambda: c.render(p, short.value),
                lambda: c.render(p, full.value + 16),                           # a pointer into the buffer: what is left counts
                lambda: c.render(api.Context.params(W, H, 2, 50, 1, flags=api.RT_FLAG_FIXED_SUMS), full.value),
                lambda: c.quantize(short.value, W * H, 2, rgb.value),
                lambda: c.quantize(full.value, W * H + 1, 2, rgb.value),
                lambda: c.resolve_sums(sums.value, W * H, short.value),
                lambda: c.render_guides(W, H, sums.value + 16),                 # two planes of float4: exactly the 32 bytes a pixel of `sums`
                lambda: c.reconstruct(short.value, sums.value, W, H, full.value),
                lambda: c.reconstruct(full.value, short.value, W, H, full.value),
                lambda: c.reconstruct(full.value, sums.value, W, H, short.value),
                lambda: c.upscale(short.value, sums.value, W, H, sums.value, W, H, full.value),
                lambda: c.upscale(full.value, sums.value, W, H, sums.value, W, H, short.value)):
        with pytest.raises(api.RtError) as e:
            bad()
        assert e.value.code == api.RT_ERR_INVALID and "bytes" in str(e.value)
    c.render(api.Context.params(W, H, 2, 50, 1, flags=api.RT_FLAG_FIXED_SUMS), sums.value)
    c.resolve_sums(sums.value, W * H, full.value)
    c.quantize(full.value, W * H, 2, rgb.value)
    c.render(p, full.value)
    c.render_guides(W, H, sums.value)
    c.reconstruct(full.value, sums.value, W, H, full.value)
    c.stats()                                                                   # waits for the stream: nothing faulted
    c.close()
    for b in (full, short, sums, rgb):
        hip.hipFree(b)
