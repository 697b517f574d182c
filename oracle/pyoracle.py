"""TEST INFRASTRUCTURE — ctypes access to the CPU checkers.

`Oracle` wraps oracle/_build/librt_oracle.so (this repo's C restatement of the reference's
per-pixel render path) and `Reference` wraps oracle/_ref/libref_main.so (the reference's own
main.cc compiled in place by oracle/Makefile).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the shipped package never does.

Scenes are passed as any object carrying the flat arrays listed in `_SCENE_FIELDS`
(raytrace_amd.scene.FlatScene has them; so does a SimpleNamespace built in a test).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "librt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_main.so")

_SCENE_FIELDS = ("prim_type", "prim_geom", "prim_mat", "mat_type", "mat_param", "mat_tex",
                 "tex_type", "tex_color", "tex_a", "tex_b", "tex_off", "texels",
                 "background", "bg_color")


def build(which=("oracle", "ref")):
    """Run the Makefile (gcc/g++ only).  `ref` is a no-op where the reference checkout is absent."""
    for target in which:
        subprocess.run(["make", "-s", "-C", _HERE, target], check=True)


class _Scene(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32), ("prim_type", C.c_void_p), ("prim_geom", C.c_void_p),
        ("prim_mat", C.c_void_p),
        ("n_mats", C.c_int32), ("mat_type", C.c_void_p), ("mat_param", C.c_void_p),
        ("mat_tex", C.c_void_p),
        ("n_tex", C.c_int32), ("tex_type", C.c_void_p), ("tex_color", C.c_void_p),
        ("tex_a", C.c_void_p), ("tex_b", C.c_void_p), ("tex_off", C.c_void_p),
        ("texels", C.c_void_p),
        ("background", C.c_int32), ("bg_color", C.c_double * 3),
        ("n_xforms", C.c_int32), ("xform", C.c_void_p), ("prim_xform", C.c_void_p),
    ]


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


class _Packed:
    """Keeps the numpy buffers alive next to the C struct that points into them."""

    def __init__(self, sc):
        self.keep = k = {}
        k["prim_type"] = _arr(sc.prim_type, np.int32)
        n = k["prim_type"].shape[0]
        k["prim_geom"] = _arr(sc.prim_geom, np.float64, (n, 6))
        k["prim_mat"] = _arr(sc.prim_mat, np.int32)
        k["mat_type"] = _arr(sc.mat_type, np.int32)
        m = k["mat_type"].shape[0]
        k["mat_param"] = _arr(sc.mat_param, np.float64, (m, 5))
        k["mat_tex"] = _arr(sc.mat_tex, np.int32)
        k["tex_type"] = _arr(sc.tex_type, np.int32)
        t = k["tex_type"].shape[0]
        k["tex_color"] = _arr(sc.tex_color, np.float64, (t, 3))
        k["tex_a"] = _arr(sc.tex_a, np.int32)
        k["tex_b"] = _arr(sc.tex_b, np.int32)
        k["tex_off"] = _arr(sc.tex_off, np.int64)
        k["texels"] = _arr(sc.texels, np.uint8)
        s = _Scene()
        s.n_prims, s.n_mats, s.n_tex = n, m, t
        for name in ("prim_type", "prim_geom", "prim_mat", "mat_type", "mat_param", "mat_tex",
                     "tex_type", "tex_color", "tex_a", "tex_b", "tex_off", "texels"):
            setattr(s, name, k[name].ctypes.data if k[name].size else None)
        xf = getattr(sc, "xform", None)
        if xf is not None and np.size(xf):
            k["xform"] = _arr(xf, np.float64).reshape(-1, 4)
            k["prim_xform"] = _arr(sc.prim_xform, np.int32)
            assert k["prim_xform"].shape[0] == n
            s.n_xforms = k["xform"].shape[0]
            s.xform = k["xform"].ctypes.data
            s.prim_xform = k["prim_xform"].ctypes.data
        s.background = int(sc.background)
        s.bg_color = (C.c_double * 3)(*[float(x) for x in sc.bg_color])
        self.struct = s

    @property
    def ref(self):
        return C.byref(self.struct)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """The C restatement.  `prec` selects the double ('f64') or float ('f32') build."""

    def __init__(self, prec="f64"):
        if not os.path.exists(ORACLE_SO):
            build(("oracle",))
        self.lib = C.CDLL(ORACLE_SO)
        self.prec = prec
        self.lib.rto_draw.restype = C.c_uint32
        self.lib.rto_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        for s in ("f64", "f32"):
            getattr(self.lib, f"rto_render_{s}").restype = C.c_uint64

    def _f(self, name):
        return getattr(self.lib, f"{name}_{self.prec}")

    # -- RNG ---------------------------------------------------------------
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.rto_philox4x32_10_kat(c, k, o)
        return [int(x) for x in o]

    def draw(self, seed, pixel, sample, segment, d):
        return int(self.lib.rto_draw(seed, pixel, sample, segment, d))

    # -- camera --------------------------------------------------------------
    def camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        out = np.zeros(22)
        a = [np.asarray(v, dtype=np.float64) for v in (lookfrom, lookat, vup)]
        self._f("rto_camera_init")(_p(a[0]), _p(a[1]), _p(a[2]), C.c_double(vfov),
                                   C.c_double(aspect), C.c_double(aperture),
                                   C.c_double(focus_dist), _p(out))
        return out

    def get_ray(self, cam, s, t, seed, pixel, sample, first_draw=2):
        ray = np.zeros(6)
        used = C.c_uint32(0)
        cam = np.ascontiguousarray(cam, dtype=np.float64)
        self._f("rto_get_ray")(_p(cam), C.c_double(s), C.c_double(t), C.c_uint64(seed),
                               C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(first_draw),
                               _p(ray), C.byref(used))
        return ray, used.value

    def sample(self, which, seed, pixel, sample, segment):
        out = np.zeros(3)
        used = C.c_uint32(0)
        idx = {"in_unit_sphere": 0, "unit_vector": 1, "in_unit_disk": 2}[which]
        self._f("rto_sample")(idx, C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample),
                              C.c_uint32(segment), _p(out), C.byref(used))
        return out, used.value

    # -- geometry --------------------------------------------------------------
    def trace(self, scene, rays, t_min=0.001, t_max=np.inf):
        pk = _Packed(scene)
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(id=np.zeros(n, np.int32), t=np.zeros(n), p=np.zeros((n, 3)),
                   n=np.zeros((n, 3)), front=np.zeros(n, np.int32), uv=np.zeros((n, 2)))
        self._f("rto_trace")(pk.ref, n, _p(rays), C.c_double(t_min), C.c_double(t_max),
                             _p(out["id"]), _p(out["t"]), _p(out["p"]), _p(out["n"]),
                             _p(out["front"]), _p(out["uv"]))
        return out

    def scatter(self, scene, mat, ray_in, p, nrm, front, uv, seed, pixel, sample, segment):
        pk = _Packed(scene)
        out = np.zeros(12)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (ray_in, p, nrm, uv)]
        ok = self._f("rto_scatter")(pk.ref, int(mat), _p(a[0]), _p(a[1]), _p(a[2]), int(front),
                                    _p(a[3]), C.c_uint64(seed), C.c_uint32(pixel),
                                    C.c_uint32(sample), C.c_uint32(segment), _p(out))
        return bool(ok), out[0:3], out[3:6], out[6:9], out[9:12]

    def texture_value(self, scene, tex, u, v, p):
        pk = _Packed(scene)
        out = np.zeros(3)
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._f("rto_texture_value")(pk.ref, int(tex), C.c_double(u), C.c_double(v), _p(p), _p(out))
        return out

    # -- integrator ------------------------------------------------------------
    def ray_color(self, scene, ray, depth, seed, pixel, sample, t_min=0.001):
        pk = _Packed(scene)
        out = np.zeros(3)
        rays = C.c_uint64(0)
        ray = np.ascontiguousarray(ray, dtype=np.float64)
        self._f("rto_ray_color")(pk.ref, _p(ray), int(depth), C.c_double(t_min), C.c_uint64(seed),
                                 C.c_uint32(pixel), C.c_uint32(sample), _p(out), C.byref(rays))
        return out, rays.value

    def render(self, scene, cam, W, H, spp, depth, seed, sample_begin=0, window=None,
               step=(1, 1), threads=1, t_min=0.001):
        """Returns (sums[H,W,3] float64 with row 0 = bottom row, rays traced)."""
        pk = _Packed(scene)
        cam = np.ascontiguousarray(cam, dtype=np.float64)
        i0, i1, j0, j1 = window if window is not None else (0, W, 0, H)
        sums = np.zeros((H, W, 3))
        rays = self._f("rto_render")(pk.ref, _p(cam), W, H, spp, sample_begin, depth,
                                     C.c_double(t_min), C.c_uint64(seed), i0, i1, j0, j1,
                                     step[0], step[1], threads, _p(sums))
        return sums, int(rays)

    def write_color(self, sums, spp):
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        out = np.zeros(sums.shape, np.int32)
        self.lib.rto_write_color(sums.size // 3, _p(sums), int(spp), _p(out))
        return out

    def camera_path_step(self, frame, state):
        state = np.ascontiguousarray(state, dtype=np.float32).copy()
        self.lib.rto_camera_path_step(int(frame), _p(state))
        return state


class Reference:
    """The reference's own main.cc, compiled by oracle/Makefile into oracle/_ref/."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = L = C.CDLL(REF_SO)
        L.ref_render_keyed.restype = C.c_uint64
        L.ref_render_libc.restype = C.c_uint64
        L.ref_ray_count.restype = C.c_uint64
        L.ref_draw_count.restype = C.c_uint64
        L.ref_draw_list_used.restype = C.c_size_t
        L.ref_set_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ref_set_draw_list.argtypes = [C.c_void_p, C.c_size_t]
        self._keep = None

    def build_default_scene(self):
        n = self.lib.ref_build_default_scene()
        out = np.zeros((n, 9))
        assert self.lib.ref_scene_dump(_p(out)) == n
        return out

    def load_scene(self, rec9):
        rec9 = np.ascontiguousarray(rec9, dtype=np.float64).reshape(-1, 9)
        assert self.lib.ref_scene_load(rec9.shape[0], _p(rec9)) == rec9.shape[0]

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        a = [np.asarray(v, dtype=np.float64) for v in (lookfrom, lookat, vup)]
        self.lib.ref_set_camera(_p(a[0]), _p(a[1]), _p(a[2]), C.c_double(vfov), C.c_double(aspect),
                                C.c_double(aperture), C.c_double(focus_dist))
        out = np.zeros(22)
        self.lib.ref_camera_dump(_p(out))
        return out

    def set_key(self, seed, pixel, sample, segment):
        self.lib.ref_set_key(seed, pixel, sample, segment)

    def set_draw_list(self, k24):
        self._keep = np.ascontiguousarray(k24, dtype=np.uint32)
        self.lib.ref_set_draw_list(_p(self._keep), self._keep.size)

    def get_ray(self, s, t):
        out = np.zeros(6)
        self.lib.ref_get_ray(C.c_double(s), C.c_double(t), _p(out))
        return out

    def sample(self, which):
        out = np.zeros(3)
        getattr(self.lib, {"in_unit_sphere": "ref_random_in_unit_sphere",
                           "unit_vector": "ref_random_unit_vector",
                           "in_unit_disk": "ref_random_in_unit_disk"}[which])(_p(out))
        return out

    def reflect(self, v, n):
        out = np.zeros(3)
        v = np.ascontiguousarray(v, dtype=np.float64); n = np.ascontiguousarray(n, dtype=np.float64)
        self.lib.ref_reflect(_p(v), _p(n), _p(out))
        return out

    def refract(self, uv, n, eta):
        out = np.zeros(3)
        uv = np.ascontiguousarray(uv, dtype=np.float64); n = np.ascontiguousarray(n, dtype=np.float64)
        self.lib.ref_refract(_p(uv), _p(n), C.c_double(eta), _p(out))
        return out

    def trace(self, rays, t_min=0.001, t_max=np.inf):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(id=np.zeros(n, np.int32), t=np.zeros(n), p=np.zeros((n, 3)),
                   n=np.zeros((n, 3)), front=np.zeros(n, np.int32))
        self.lib.ref_trace(n, _p(rays), C.c_double(t_min), C.c_double(t_max), _p(out["id"]),
                           _p(out["t"]), _p(out["p"]), _p(out["n"]), _p(out["front"]))
        return out

    def scatter(self, obj, ray_in, p, nrm, front):
        out = np.zeros(9)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (ray_in, p, nrm)]
        ok = self.lib.ref_scatter(int(obj), _p(a[0]), _p(a[1]), _p(a[2]), int(front), _p(out))
        return bool(ok), out[0:3], out[3:6], out[6:9]

    def ray_color(self, ray, depth, seed, pixel, sample):
        out = np.zeros(3)
        ray = np.ascontiguousarray(ray, dtype=np.float64)
        r0 = self.lib.ref_ray_count()
        self.lib.ref_ray_color(_p(ray), int(depth), C.c_uint64(seed), C.c_uint32(pixel),
                               C.c_uint32(sample), _p(out))
        return out, int(self.lib.ref_ray_count() - r0)

    def render_keyed(self, W, H, spp, depth, seed, sample_begin=0, window=None, step=(1, 1)):
        i0, i1, j0, j1 = window if window is not None else (0, W, 0, H)
        sums = np.zeros((H, W, 3))
        rays = self.lib.ref_render_keyed(W, H, spp, sample_begin, depth, C.c_uint64(seed),
                                         i0, i1, j0, j1, step[0], step[1], _p(sums))
        return sums, int(rays)

    def render_libc(self, W, H, spp, depth, window=None, step=(1, 1), want_sums=False):
        i0, i1, j0, j1 = window if window is not None else (0, W, 0, H)
        sums = np.zeros((H, W, 3)) if want_sums else None
        rays = self.lib.ref_render_libc(W, H, spp, depth, i0, i1, j0, j1, step[0], step[1],
                                        _p(sums) if want_sums else None)
        return sums, int(rays)

    def write_color(self, sums, spp):
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        out = np.zeros(sums.shape, np.int32)
        assert self.lib.ref_write_color(sums.size // 3, _p(sums), int(spp), _p(out)) == sums.size // 3
        return out
