"""ctypes binding of librt_amd.so (include/rt.h).  Arrays cross the boundary as plain pointers;
device images are addressed by integer pointers (e.g. torch.Tensor.data_ptr())."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_NO_SCENE, RT_ERR_NO_CAMERA, RT_ERR_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6
RT_ACCEL_AUTO, RT_ACCEL_LIST, RT_ACCEL_BVH, RT_ACCEL_BVH_DEVICE = 0, 1, 2, 3
TRACE_NOALLOC_MAX_RAYS = 65536           # RT_TRACE_NOALLOC_MAX_RAYS
RT_FLAG_MEAN, RT_FLAG_NO_REFINE, RT_FLAG_COUNTERS, RT_FLAG_PHASE_CLOCKS, RT_FLAG_NO_LDS_SCENE, RT_FLAG_FIXED_SUMS, RT_FLAG_FLOAT_NODES = 1, 2, 4, 8, 32, 64, 128
RT_FLAG_ALL_FEATURES = 256

EXPORTS = ["rt_create", "rt_destroy", "rt_set_stream", "rt_upload_scene", "rt_set_camera",
           "rt_shard_rows", "rt_render", "rt_render_host", "rt_trace_primary", "rt_trace_primary_with", "rt_trace_primary_noalloc", "rt_quantize",
           "rt_resolve_sums", "rt_render_guides", "rt_reconstruct", "rt_upscale", "rt_get_stats", "rt_last_error", "rt_version",
           "rt_decode_jpeg", "rt_load_jpeg_file", "rt_free_image", "rt_image_last_error", "rt_write_ppm"]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_prims", C.c_int32),
        ("prim_type", C.c_void_p), ("prim_geom", C.c_void_p), ("prim_mat", C.c_void_p),
        ("n_mats", C.c_int32), ("mat_type", C.c_void_p), ("mat_param", C.c_void_p), ("mat_tex", C.c_void_p),
        ("n_tex", C.c_int32), ("tex_type", C.c_void_p), ("tex_color", C.c_void_p), ("tex_a", C.c_void_p),
        ("tex_b", C.c_void_p), ("tex_off", C.c_void_p), ("texels", C.c_void_p), ("texel_bytes", C.c_uint64),
        ("background", C.c_int32), ("bg_color", C.c_float * 3), ("accel", C.c_int32),
        ("wide_radius", C.c_double),
        ("n_xforms", C.c_int32), ("xform", C.c_void_p), ("prim_xform", C.c_void_p),
    ]


class CameraDesc(C.Structure):
    _fields_ = [("lookfrom", C.c_double * 3), ("lookat", C.c_double * 3), ("vup", C.c_double * 3),
                ("vfov", C.c_double), ("aspect", C.c_double), ("aperture", C.c_double),
                ("focus_dist", C.c_double)]


class RenderParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
                ("spp", C.c_int32), ("sample_begin", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("tile_rows", C.c_int32), ("tile_first", C.c_int32),
                ("tile_step", C.c_int32), ("t_min", C.c_float), ("flags", C.c_uint32),
                ("lanes_per_pixel", C.c_int32), ("traversal_slice", C.c_int32), ("leaf_quorum", C.c_int32),
                ("stack_lds_entries", C.c_int32), ("traversal_exit_share", C.c_int32)]


class ReconstructParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("iterations", C.c_int32), ("sigma_color", C.c_float),
                ("sigma_normal", C.c_float), ("sigma_depth", C.c_float), ("albedo_floor", C.c_float), ("flags", C.c_uint32)]


RT_RECON_NO_LDS = 1


class Hit(C.Structure):
    _fields_ = [("id", C.c_int32), ("t", C.c_float), ("p", C.c_float * 3), ("n", C.c_float * 3),
                ("front_face", C.c_int32), ("u", C.c_float), ("v", C.c_float)]


HIT_DTYPE = np.dtype([("id", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("n", "<f4", 3),
                      ("front_face", "<i4"), ("u", "<f4"), ("v", "<f4")])


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("rays", C.c_uint64), ("samples", C.c_uint64), ("box_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("sphere_hits", C.c_uint64), ("rect_tests", C.c_uint64),
                ("wide_tests", C.c_uint64), ("shades", C.c_uint64), ("phase_clocks", C.c_uint64 * 5),
                ("kernel_ms", C.c_float),
                ("lanes_per_pixel", C.c_int32), ("accel", C.c_int32), ("bvh_nodes", C.c_int32),
                ("bvh_depth", C.c_int32), ("build_ms", C.c_float), ("bvh_coarse_nodes", C.c_int32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["phase_clocks"] = [int(x) for x in self.phase_clocks]
        return d


_lib = None


def lib():
    """Loads librt_amd.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytrace_amd/csrc` "
                              "(or __graft_entry__.build())")
        # When PyTorch shares the process it must load its own HIP runtime first: two copies of
        # libamdhip64 cannot both initialise the GPU, and the first one loaded wins the soname.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rt_destroy.argtypes = [C.c_void_p]
        L.rt_destroy.restype = None
        L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rt_set_camera.argtypes = [C.c_void_p, C.POINTER(CameraDesc)]
        L.rt_shard_rows.argtypes = [C.c_int32] * 4
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
        L.rt_render_host.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
        L.rt_trace_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
        L.rt_trace_primary_with.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_uint32, C.c_void_p]
        L.rt_trace_primary_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
        L.rt_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        L.rt_resolve_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        L.rt_render_guides.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.rt_reconstruct.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ReconstructParams), C.c_void_p]
        L.rt_upscale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                 C.POINTER(ReconstructParams), C.c_void_p]
        L.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_last_error.restype = C.c_char_p
        L.rt_version.restype = C.c_char_p
        L.rt_decode_jpeg.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.rt_load_jpeg_file.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.rt_free_image.argtypes = [C.c_void_p]
        L.rt_free_image.restype = None
        L.rt_image_last_error.restype = C.c_char_p
        L.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        _lib = L
    return _lib


def _take_image(rc, w, h, out):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_image_last_error().decode())
    try:
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (h.value, w.value, 3)).copy()
    finally:
        lib().rt_free_image(out)


def decode_jpeg(data: bytes) -> np.ndarray:
    """JPEG file image in memory -> uint8 [height, width, 3], top row first (rt_decode_jpeg)."""
    w, h, out = C.c_int32(), C.c_int32(), C.c_void_p()
    data = bytes(data)
    return _take_image(lib().rt_decode_jpeg(data, len(data), C.byref(w), C.byref(h), C.byref(out)), w, h, out)


def load_jpeg_file(path) -> np.ndarray:
    w, h, out = C.c_int32(), C.c_int32(), C.c_void_p()
    return _take_image(lib().rt_load_jpeg_file(os.fsencode(path), C.byref(w), C.byref(h), C.byref(out)), w, h, out)


def write_ppm(path, rgb8_bottom_up, binary=False) -> None:
    """uint8 [height, width, 3] with row 0 = bottom row -> a P3 (text) or P6 file (rt_write_ppm)."""
    a = np.ascontiguousarray(rgb8_bottom_up, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_ppm takes [height, width, 3] bytes")
    rc = lib().rt_write_ppm(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], 1 if binary else 0)
    if rc != RT_OK:
        raise RtError(rc, lib().rt_image_last_error().decode())


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


def pack_scene(flat, accel=RT_ACCEL_AUTO, wide_radius=0.0):
    """FlatScene (raytrace_amd.scene) -> (SceneDesc, keepalive list)."""
    k = dict(
        prim_type=np.ascontiguousarray(flat.prim_type, np.int32),
        prim_geom=np.ascontiguousarray(flat.prim_geom, np.float64),
        prim_mat=np.ascontiguousarray(flat.prim_mat, np.int32),
        mat_type=np.ascontiguousarray(flat.mat_type, np.int32),
        mat_param=np.ascontiguousarray(flat.mat_param, np.float64),
        mat_tex=np.ascontiguousarray(flat.mat_tex, np.int32),
        tex_type=np.ascontiguousarray(flat.tex_type, np.int32),
        tex_color=np.ascontiguousarray(flat.tex_color, np.float64),
        tex_a=np.ascontiguousarray(flat.tex_a, np.int32),
        tex_b=np.ascontiguousarray(flat.tex_b, np.int32),
        tex_off=np.ascontiguousarray(flat.tex_off, np.int64),
        texels=np.ascontiguousarray(flat.texels, np.uint8),
    )
    d = SceneDesc()
    d.struct_size = C.sizeof(SceneDesc)
    d.n_prims, d.n_mats, d.n_tex = k["prim_type"].shape[0], k["mat_type"].shape[0], k["tex_type"].shape[0]
    for name, arr in k.items():
        setattr(d, name, _ptr(arr))
    d.texel_bytes = k["texels"].size
    d.background = int(flat.background)
    d.bg_color = (C.c_float * 3)(*[float(x) for x in flat.bg_color])
    d.accel = int(accel)
    d.wide_radius = float(wide_radius)
    xf = getattr(flat, "xform", None)
    if xf is not None and np.size(xf):
        k["xform"] = np.ascontiguousarray(xf, np.float64).reshape(-1, 4)
        k["prim_xform"] = np.ascontiguousarray(flat.prim_xform, np.int32)
        if k["prim_xform"].shape[0] != d.n_prims:
            raise ValueError("prim_xform needs one entry per primitive")
        d.n_xforms = k["xform"].shape[0]
        d.xform, d.prim_xform = _ptr(k["xform"]), _ptr(k["prim_xform"])
    return d, k


class Context:
    """One GPU's render context (rt_context)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().rt_create(int(device), C.byref(self._h))
        if rc != RT_OK:
            raise RtError(rc, "rt_create failed (no gfx950 device %d?)" % device)
        self.device = device

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, lib().rt_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        self._check(lib().rt_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def upload_scene(self, flat, accel=RT_ACCEL_AUTO, wide_radius=0.0):
        d, keep = pack_scene(flat, accel, wide_radius)
        self._check(lib().rt_upload_scene(self._h, C.byref(d)))
        del keep

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        c = CameraDesc()
        c.lookfrom = (C.c_double * 3)(*[float(x) for x in lookfrom])
        c.lookat = (C.c_double * 3)(*[float(x) for x in lookat])
        c.vup = (C.c_double * 3)(*[float(x) for x in vup])
        c.vfov, c.aspect, c.aperture, c.focus_dist = float(vfov), float(aspect), float(aperture), float(focus_dist)
        self._check(lib().rt_set_camera(self._h, C.byref(c)))

    @staticmethod
    def params(width, height, spp, max_depth=50, seed=1, sample_begin=0, tiles=(0, 0, 0), t_min=0.0,
               flags=0, lanes_per_pixel=0, traversal_slice=0, leaf_quorum=0, stack_lds_entries=0, traversal_exit_share=0):
        p = RenderParams()
        p.struct_size = C.sizeof(RenderParams)
        p.width, p.height, p.spp, p.sample_begin, p.max_depth = width, height, spp, sample_begin, max_depth
        p.seed = seed
        p.tile_rows, p.tile_first, p.tile_step = tiles
        p.t_min, p.flags, p.lanes_per_pixel = t_min, flags, lanes_per_pixel
        p.traversal_slice, p.leaf_quorum, p.stack_lds_entries = traversal_slice, leaf_quorum, stack_lds_entries
        p.traversal_exit_share = traversal_exit_share
        return p

    @staticmethod
    def shard_rows(height, tiles=(0, 0, 0)):
        r = lib().rt_shard_rows(height, *tiles)
        if r < 0:
            raise RtError(r, "bad tile sharding")
        return r

    def render(self, params, device_ptr):
        """Asynchronous render into device memory (float4 per pixel, packed shard rows)."""
        self._check(lib().rt_render(self._h, C.byref(params), C.c_void_p(device_ptr)))

    def render_host(self, params):
        rows = self.shard_rows(params.height, (params.tile_rows, params.tile_first, params.tile_step))
        # RT_FLAG_FIXED_SUMS: (r, g, b sums in 32.32 fixed point, samples) as uint64 per pixel
        out = np.empty((rows, params.width, 4), np.uint64 if params.flags & RT_FLAG_FIXED_SUMS else np.float32)
        self._check(lib().rt_render_host(self._h, C.byref(params), out.ctypes.data))
        return out

    def trace_primary(self, rays, t_min=0.001, noalloc=False, walk=None):
        """noalloc: through rt_trace_primary_noalloc (fixed device staging, at most
        TRACE_NOALLOC_MAX_RAYS rays, no device allocation in the call).
        walk: through rt_trace_primary_with: 0 = the walk a frame of the scene uses (LDS where it fits), or
        RT_FLAG_NO_LDS_SCENE / RT_FLAG_FLOAT_NODES."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        if walk is not None:
            self._check(lib().rt_trace_primary_with(self._h, rays.ctypes.data, rays.shape[0], C.c_float(t_min), int(walk), hits.ctypes.data))
            return hits
        fn = lib().rt_trace_primary_noalloc if noalloc else lib().rt_trace_primary
        self._check(fn(self._h, rays.ctypes.data, rays.shape[0], C.c_float(t_min), hits.ctypes.data))
        return hits

    def quantize(self, rgba_device_ptr, n_pixels, spp, rgb8_device_ptr):
        self._check(lib().rt_quantize(self._h, C.c_void_p(rgba_device_ptr), n_pixels, spp,
                                      C.c_void_p(rgb8_device_ptr)))

    def resolve_sums(self, sums_device_ptr, n_pixels, rgba_device_ptr, mean=False):
        """RT_FLAG_FIXED_SUMS image (device) -> float4 image (device), as rt_render converts its own sums."""
        self._check(lib().rt_resolve_sums(self._h, C.c_void_p(sums_device_ptr), n_pixels, int(bool(mean)),
                                          C.c_void_p(rgba_device_ptr)))

    def render_guides(self, width, height, guides_device_ptr):
        """First-hit guides of the reconstruction stage into device memory: two planes of height x width
        float4 (normal + distance; albedo + list position as int32 bits)."""
        self._check(lib().rt_render_guides(self._h, width, height, C.c_void_p(guides_device_ptr)))

    def reconstruct(self, rgba_device_ptr, guides_device_ptr, width, height, out_device_ptr, iterations=0, sigma_color=0.0,
                    sigma_normal=0.0, sigma_depth=0.0, albedo_floor=0.0, flags=0):
        """Edge-avoiding a-trous reconstruction of a low-sample-count frame (rt_reconstruct); 0 = default."""
        p = ReconstructParams()
        p.struct_size = C.sizeof(ReconstructParams)
        p.iterations, p.sigma_color, p.sigma_normal, p.sigma_depth, p.albedo_floor = iterations, sigma_color, sigma_normal, sigma_depth, albedo_floor
        p.flags = flags
        self._check(lib().rt_reconstruct(self._h, C.c_void_p(rgba_device_ptr), C.c_void_p(guides_device_ptr), width, height,
                                         C.byref(p), C.c_void_p(out_device_ptr)))

    def upscale(self, rgba_low_ptr, guides_low_ptr, low_width, low_height, guides_high_ptr, width, height, out_device_ptr,
                sigma_normal=0.0, sigma_depth=0.0, albedo_floor=0.0):
        """Joint bilateral upsampling of a low-resolution frame onto the grid of high-resolution guides (rt_upscale)."""
        p = ReconstructParams()
        p.struct_size = C.sizeof(ReconstructParams)
        p.sigma_normal, p.sigma_depth, p.albedo_floor = sigma_normal, sigma_depth, albedo_floor
        self._check(lib().rt_upscale(self._h, C.c_void_p(rgba_low_ptr), C.c_void_p(guides_low_ptr), low_width, low_height,
                                     C.c_void_p(guides_high_ptr), width, height, C.byref(p), C.c_void_p(out_device_ptr)))

    def stats(self):
        s = Stats()
        s.struct_size = C.sizeof(Stats)
        self._check(lib().rt_get_stats(self._h, C.byref(s)))
        return s.as_dict()
