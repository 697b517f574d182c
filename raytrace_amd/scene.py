"""Host-side scene vocabulary of the reference, flattened for the GPU path.

The class names, constructor argument order and meaning follow the reference's scene API so
that scene-authoring code reads the same:

    sphere(center, radius, material)                      sphere.h:13, cuda/sphere.h:16
    xy_rect(x0,x1,y0,y1,k,mat) / xz_rect / yz_rect        cuda/rect.h:19,66,112
    sceneobject_list.add / clear                          sceneobject_list.h:16-21
    bvh_node(list)                                        cuda/bvh.h:18
    lambertian(color|texture) metal(color,fuzz) dielectric(ir)
                                                          material.h:19,40,56
    lambertian_bg(color|texture) diffuse_light(color|texture)
                                                          cuda/material.h:37-38,107-109
    rgb_color(color) checker_texture(c1,c2 | even,odd) image_texture(width,height,texels)
                                                          cuda/texture.h:22,39-41,61
    camera(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist)
                                                          camera.h:19-20

Nothing here touches the GPU: `flatten()` turns a sceneobject_list into the flat arrays that
`rt_upload_scene` takes (see include/rt.h), in list order, de-duplicating materials and
textures by object identity.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np

# primitive / material / texture / background tags: the values of include/rt.h
RT_SPHERE, RT_XY_RECT, RT_XZ_RECT, RT_YZ_RECT = 0, 1, 2, 3
RT_LAMBERTIAN, RT_METAL, RT_DIELECTRIC, RT_DIFFUSE_LIGHT, RT_LAMBERTIAN_BG = 0, 1, 2, 3, 4
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE = 0, 1, 2
RT_BG_SKY, RT_BG_SOLID = 0, 1


def vec3(x=0.0, y=0.0, z=0.0):
    return (float(x), float(y), float(z))


point3 = vec3
color = vec3


def _v(a):
    a = tuple(float(x) for x in a)
    if len(a) != 3:
        raise ValueError("expected 3 components")
    return a


# ---- textures ------------------------------------------------------------------------

class custom_texture:
    pass


class rgb_color(custom_texture):
    def __init__(self, *c):
        self.color_value = _v(c[0] if len(c) == 1 else c)


class checker_texture(custom_texture):
    def __init__(self, even, odd):
        self.even = even if isinstance(even, custom_texture) else rgb_color(even)
        self.odd = odd if isinstance(odd, custom_texture) else rgb_color(odd)


class image_texture(custom_texture):
    """texels: uint8 array [height, width, 3], top row first (what the reference's loader
    hands to the device, cuda/main.cu:393-406)."""

    def __init__(self, width, height, texels):
        texels = np.ascontiguousarray(texels, dtype=np.uint8)
        if texels.shape != (height, width, 3):
            raise ValueError(f"texels must be [{height},{width},3] uint8, got {texels.shape}")
        self.texture_width, self.texture_height, self.texels = int(width), int(height), texels


def load_texture(filename):
    """load_texture of cuda/main.cu:385-436: a JPEG file -> image_texture.  Decoding is done by
    the library (rt_load_jpeg_file), which reproduces the texels of the reference's stb_image."""
    from . import api
    texels = api.load_jpeg_file(filename)
    return image_texture(texels.shape[1], texels.shape[0], texels)


# ---- materials --------------------------------------------------------------------------

class material:
    pass


def _tex_or_color(a):
    return a if isinstance(a, custom_texture) else _v(a)


class lambertian(material):
    def __init__(self, a):
        self.albedo = _tex_or_color(a)


class lambertian_bg(material):
    def __init__(self, a):
        self.albedo = _tex_or_color(a)


class metal(material):
    def __init__(self, a, f):
        self.albedo = _v(a)
        self.fuzz = float(f) if f < 1 else 1.0        # material.h:40


class dielectric(material):
    def __init__(self, index_of_refraction):
        self.ir = float(index_of_refraction)


class diffuse_light(material):
    def __init__(self, a):
        self.emit = _tex_or_color(a)


# ---- objects ----------------------------------------------------------------------------------

class sceneobject:
    pass


class sphere(sceneobject):
    def __init__(self, cen, r, m):
        self.center, self.radius, self.mat_ptr = _v(cen), float(r), m


class _rect(sceneobject):
    kind = None

    def __init__(self, a0, a1, b0, b1, k, mat):
        self.a0, self.a1, self.b0, self.b1, self.k, self.mat = (float(a0), float(a1), float(b0),
                                                               float(b1), float(k), mat)


class xy_rect(_rect):
    kind = RT_XY_RECT


class xz_rect(_rect):
    kind = RT_XZ_RECT


class yz_rect(_rect):
    kind = RT_YZ_RECT


class sceneobject_list(sceneobject):
    def __init__(self, obj=None):
        self.objects = []
        if obj is not None:
            if isinstance(obj, (list, tuple)):
                self.objects.extend(obj)
            else:
                self.add(obj)

    def clear(self):
        self.objects.clear()

    def add(self, obj):
        self.objects.append(obj)


class rotate_y(sceneobject):
    """cuda/sceneobject.h:31-47: `p` turned by `angle` degrees about the world's +y axis."""

    def __init__(self, p, angle):
        self.ptr, self.angle = p, float(angle)


class translate(sceneobject):
    """`p` moved by `displacement`.  The reference's tree only has the rotation wrapper; the
    translation is the companion every use of rotate_y needs (an object turns about the world's
    y axis, not its own)."""

    def __init__(self, p, displacement):
        self.ptr, self.offset = p, _v(displacement)


def box(p0, p1, mat, mat_low=None):
    """The six rects of an axis-aligned box in the order of cuda/main.cu:130-137: per axis pair
    the face at p1 first (material `mat`), then the face at p0 (`mat_low`, default `mat`)."""
    p0, p1 = _v(p0), _v(p1)
    lo = mat if mat_low is None else mat_low
    return sceneobject_list([
        xy_rect(p0[0], p1[0], p0[1], p1[1], p1[2], mat), xy_rect(p0[0], p1[0], p0[1], p1[1], p0[2], lo),
        xz_rect(p0[0], p1[0], p0[2], p1[2], p1[1], mat), xz_rect(p0[0], p1[0], p0[2], p1[2], p0[1], lo),
        yz_rect(p0[1], p1[1], p0[2], p1[2], p1[0], mat), yz_rect(p0[1], p1[1], p0[2], p1[2], p0[0], lo)])


class bvh_node(sceneobject):
    """Grouping marker kept for source compatibility (cuda/bvh.h:18).  The GPU path builds one
    BVH over the whole flattened scene, so a bvh_node only contributes its members."""

    def __init__(self, scene_list, rng=None):
        self.objects = list(scene_list.objects)


# ---- camera ------------------------------------------------------------------------------------

class camera:
    """camera.h:19-38; the optional `angles` and the mutators are the CUDA tree's
    (cuda/camera.h:29-117) and work in float32 like the original."""

    def __init__(self, lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist,
                 angles=(0.0, 0.0, 0.0)):
        self.look_from, self.look_at, self.vup = _v(lookfrom), _v(lookat), _v(vup)
        self.angles = _v(angles)
        self.vfov, self.aspect_ratio = float(vfov), float(aspect_ratio)
        self.aperture, self.focus_dist = float(aperture), float(focus_dist)

    def desc(self):
        return dict(lookfrom=self.look_from, lookat=self.look_at, vup=self.vup, vfov=self.vfov,
                    aspect=self.aspect_ratio, aperture=self.aperture, focus_dist=self.focus_dist)

    def compute_camera_scene(self):
        """No cached basis on the host: rt_set_camera derives it (include/rt.h)."""
        return self.desc()

    def translate_camera(self, x, y, z):                  # cuda/camera.h:59-63
        f = np.float32
        lf = [f(c) for c in self.look_from]
        self.look_from = (float(lf[0] + f(x)), float(lf[1] + f(y)), float(lf[2] + f(z)))

    @staticmethod
    def _rads(deg):                                        # cuda/utils.h:17
        f = np.float32
        return f(f(deg) * f(3.1415926535897932385) / f(180.0))

    def _rot(self, axis_angle, new_angle, i, j, sign):
        f = np.float32
        r = self._rads(f(new_angle) - f(self.angles[axis_angle]))
        c, s = f(math.cos(float(r))), f(math.sin(float(r)))
        # float cos/sin of a float argument, rounded to float like cosf/sinf
        lf = [f(v) for v in self.look_from]
        ni = f(lf[i] * c) - f(sign) * f(lf[j] * s)
        nj = f(sign) * f(lf[i] * s) + f(lf[j] * c)
        lf[i], lf[j] = f(ni), f(nj)
        self.look_from = tuple(float(v) for v in lf)
        a = list(self.angles)
        a[axis_angle] = float(f(new_angle))
        self.angles = tuple(a)

    def rotate_camera_x(self, new_x_angle):               # cuda/camera.h:69-81 (xy-plane)
        self._rot(0, new_x_angle, 0, 1, 1.0)

    def rotate_camera_y(self, new_y_angle):               # cuda/camera.h:87-99 (xz-plane)
        f = np.float32
        r = self._rads(f(new_y_angle) - f(self.angles[1]))
        c, s = f(math.cos(float(r))), f(math.sin(float(r)))
        x, y, z = (f(v) for v in self.look_from)
        nx = f(f(x * c) + f(z * s))
        nz = f(f(z * c) - f(x * s))
        self.look_from = (float(nx), float(y), float(nz))
        self.angles = (self.angles[0], float(f(new_y_angle)), self.angles[2])

    def rotate_camera_z(self, new_z_angle):               # cuda/camera.h:105-117 (yz-plane)
        self._rot(2, new_z_angle, 1, 2, 1.0)


def update_scene(cam: camera, frame: int):
    """Per-frame camera animation of cuda/main.cu:220-279 (rotate_scene: :206-217)."""
    f = np.float32

    def rotate(ax, ay, az):
        if ax != cam.angles[0]:
            cam.rotate_camera_x(ax)
        if ay != cam.angles[1]:
            cam.rotate_camera_y(ay)
        if az != cam.angles[2]:
            cam.rotate_camera_z(az)

    ax, ay, az = (f(a) for a in cam.angles)
    d = f(0.125)
    if frame == 0:
        return
    if frame < 200:
        rotate(float(ax), float(f(ay + d)), float(az))
    elif frame < 400:
        rotate(float(ax), float(f(ay - d)), float(az))
    elif frame < 500:
        rotate(float(f(ax + d)), float(ay), float(az))
        cam.translate_camera(-0.025, 0.025, 0.0)
    elif frame < 600:
        rotate(float(f(ax - d)), float(ay), float(az))
        cam.translate_camera(0.025, -0.025, 0.0)


# ---- flattening -------------------------------------------------------------------------------------

@dataclass
class FlatScene:
    prim_type: np.ndarray
    prim_geom: np.ndarray      # [n,6] float64: sphere cx cy cz r 0 0 | rect a0 a1 b0 b1 k 0
    prim_mat: np.ndarray
    mat_type: np.ndarray
    mat_param: np.ndarray      # [m,5] float64: albedo rgb, fuzz, ir
    mat_tex: np.ndarray
    tex_type: np.ndarray
    tex_color: np.ndarray      # [t,3] float64
    tex_a: np.ndarray
    tex_b: np.ndarray
    tex_off: np.ndarray
    texels: np.ndarray         # uint8 blob
    background: int = RT_BG_SKY
    bg_color: tuple = (0.0, 0.0, 0.0)
    meta: dict = field(default_factory=dict)
    xform: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float64))   # [x,4] degrees about +y, offset
    prim_xform: np.ndarray = None                                                     # [n] index into xform or -1

    @property
    def n_prims(self):
        return int(self.prim_type.shape[0])


def _walk(obj, out, xf=None):
    """Collects (primitive, transform) leaves.  A transform is (angle, offset) meaning
    world = R_y(angle) * object + offset; wrappers compose into that one form:
    translate adds to the offset, rotate_y adds to the angle and turns the offset."""
    if isinstance(obj, (sceneobject_list, bvh_node)):
        for o in obj.objects:
            _walk(o, out, xf)
    elif isinstance(obj, (translate, rotate_y)):
        # the wrapper acts on what its child produces, so the child's own transform is the
        # inner one: walk down with a list of wrappers, outermost first
        _walk(obj.ptr, out, (xf or ()) + (obj,))
    elif isinstance(obj, (sphere, _rect)):
        out.append((obj, _compose(xf)))
    else:
        raise TypeError(f"not a scene object: {type(obj).__name__}")


def _compose(wrappers):
    if not wrappers:
        return None
    angle, off = 0.0, np.zeros(3)
    for w in reversed(wrappers):                      # innermost wrapper acts first
        if isinstance(w, translate):
            off = off + w.offset
        else:
            r = np.deg2rad(w.angle)
            c, s_ = np.cos(r), np.sin(r)
            off = np.array([c * off[0] + s_ * off[2], off[1], -s_ * off[0] + c * off[2]])   # cuda/sceneobject.h:98-99
            angle += w.angle
    return (angle, *off)


def flatten(world, background=RT_BG_SKY, bg_color=(0.0, 0.0, 0.0)) -> FlatScene:
    leaves = []
    _walk(world, leaves)
    prims = [o for o, _ in leaves]
    xforms, xform_ids = [], {}
    pxf = np.full(len(leaves), -1, np.int32)
    for i, (_, xf) in enumerate(leaves):
        if xf is not None:
            if xf not in xform_ids:
                xform_ids[xf] = len(xforms)
                xforms.append(xf)
            pxf[i] = xform_ids[xf]
    mats, mat_ids = [], {}
    texs, tex_ids = [], {}
    blobs, blob_bytes = [], 0

    def tex_id(t):
        nonlocal blob_bytes
        if id(t) in tex_ids:
            return tex_ids[id(t)]
        if isinstance(t, rgb_color):
            rec = (RT_TEX_SOLID, t.color_value, 0, 0, 0)
        elif isinstance(t, checker_texture):
            rec = (RT_TEX_CHECKER, (0, 0, 0), tex_id(t.even), tex_id(t.odd), 0)
        elif isinstance(t, image_texture):
            rec = (RT_TEX_IMAGE, (0, 0, 0), t.texture_width, t.texture_height, blob_bytes)
            blobs.append(t.texels.reshape(-1))
            blob_bytes += t.texels.size
        else:
            raise TypeError(f"not a texture: {type(t).__name__}")
        tex_ids[id(t)] = len(texs)
        texs.append(rec)
        return tex_ids[id(t)]

    def mat_id(m):
        if id(m) in mat_ids:
            return mat_ids[id(m)]
        tex, alb, fuzz, ir = -1, (0.0, 0.0, 0.0), 0.0, 0.0
        if isinstance(m, lambertian):
            typ, a = RT_LAMBERTIAN, m.albedo
        elif isinstance(m, lambertian_bg):
            typ, a = RT_LAMBERTIAN_BG, m.albedo
        elif isinstance(m, diffuse_light):
            typ, a = RT_DIFFUSE_LIGHT, m.emit
        elif isinstance(m, metal):
            typ, a, fuzz = RT_METAL, m.albedo, m.fuzz
        elif isinstance(m, dielectric):
            typ, a, ir = RT_DIELECTRIC, (1.0, 1.0, 1.0), m.ir
        else:
            raise TypeError(f"not a material: {type(m).__name__}")
        if isinstance(a, custom_texture):
            tex = tex_id(a)
        else:
            alb = a
        mat_ids[id(m)] = len(mats)
        mats.append((typ, alb, fuzz, ir, tex))
        return mat_ids[id(m)]

    n = len(prims)
    ptype = np.zeros(n, np.int32)
    geom = np.zeros((n, 6), np.float64)
    pmat = np.zeros(n, np.int32)
    for i, o in enumerate(prims):
        if isinstance(o, sphere):
            ptype[i] = RT_SPHERE
            geom[i, :4] = (*o.center, o.radius)
            pmat[i] = mat_id(o.mat_ptr)
        else:
            ptype[i] = o.kind
            geom[i, :5] = (o.a0, o.a1, o.b0, o.b1, o.k)
            pmat[i] = mat_id(o.mat)
    m = len(mats)
    t = len(texs)
    return FlatScene(
        prim_type=ptype, prim_geom=geom, prim_mat=pmat,
        mat_type=np.array([x[0] for x in mats], np.int32).reshape(m),
        mat_param=np.array([[*x[1], x[2], x[3]] for x in mats], np.float64).reshape(m, 5),
        mat_tex=np.array([x[4] for x in mats], np.int32).reshape(m),
        tex_type=np.array([x[0] for x in texs], np.int32).reshape(t),
        tex_color=np.array([x[1] for x in texs], np.float64).reshape(t, 3),
        tex_a=np.array([x[2] for x in texs], np.int32).reshape(t),
        tex_b=np.array([x[3] for x in texs], np.int32).reshape(t),
        tex_off=np.array([x[4] for x in texs], np.int64).reshape(t),
        texels=(np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)),
        background=int(background), bg_color=_v(bg_color),
        xform=np.array(xforms, np.float64).reshape(len(xforms), 4), prim_xform=pxf)


# ---- scenes of the benchmark configurations ----------------------------------------------------------

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def rtiow_records():
    """The reference's default scene (main.cc:37-85) as its g++ build produces it:
    487 x (cx cy cz r type albedo(3) fuzz|ir).  The scene depends on the compiler's argument
    evaluation order (main.cc:47,54), so it is stored, never re-derived."""
    return np.load(os.path.join(_DATA, "rtiow_gxx_scene.npy"))


def world_from_records(rec):
    world = sceneobject_list()
    for o in np.asarray(rec, dtype=np.float64).reshape(-1, 9):
        typ = int(o[4])
        if typ == 0:
            m = lambertian(o[5:8])
        elif typ == 1:
            m = metal(o[5:8], o[8])
        elif typ == 2:
            m = dielectric(o[8])
        else:
            raise ValueError(f"unknown material tag {typ}")
        world.add(sphere(o[0:3], o[3], m))
    return world


def rtiow_scene():
    """main.cc's default world and camera (main.cc:97-106); aspect is the caller's."""
    return world_from_records(rtiow_records())


def rtiow_camera(aspect_ratio):
    return camera((6, 3, -13), (0, 0, 0), (0, 1, 0), 20, aspect_ratio, 0.1, 10.0)


def cornell_scene():
    """The reference's 555-unit room (cuda/main.cu:162-170) with the lambertian walls of
    cuda/main.cu:81-83 in place of the skybox images, the light of cuda/main.cu:67,166 and
    the glass and metal r=30 spheres of cuda/main.cu:99-100.  Background is black."""
    white = lambertian((.73, .73, .73))
    red = lambertian((.65, .05, .05))
    green = lambertian((.12, .45, .15))
    light = diffuse_light((7, 7, 7))
    w = sceneobject_list()
    w.add(sphere((200, 278, 400), 30.0, dielectric(1.5)))
    w.add(sphere((100, 278, 350), 30.0, metal((0.7, 0.6, 0.5), 0.0)))
    w.add(yz_rect(0, 555, 0, 555, 555, green))
    w.add(yz_rect(0, 555, 0, 555, 0, red))
    w.add(xz_rect(75, 270, 76, 280, 554, light))
    w.add(xz_rect(0, 555, 0, 555, 0, white))
    w.add(xz_rect(0, 555, 0, 555, 555, white))
    w.add(xy_rect(0, 555, 0, 555, 555, white))
    return w


def cornell_boxes_scene():
    """The same room holding two turned boxes instead of the spheres: the arrangement
    cuda/sceneobject.h's rotate_y exists for (a box of six rects, cuda/main.cu:126-137, turned
    about y and set in place).  Sizes and angles: 165 x 330 x 165 at +15 degrees moved to
    (265, 0, 295); 165 x 165 x 165 at -18 degrees moved to (130, 0, 65).  A glass sphere rides on
    the short box so that a refracting object is seen through an instance."""
    white = lambertian((.73, .73, .73))
    red = lambertian((.65, .05, .05))
    green = lambertian((.12, .45, .15))
    light = diffuse_light((7, 7, 7))
    w = sceneobject_list()
    w.add(yz_rect(0, 555, 0, 555, 555, green))
    w.add(yz_rect(0, 555, 0, 555, 0, red))
    w.add(xz_rect(75, 270, 76, 280, 554, light))
    w.add(xz_rect(0, 555, 0, 555, 0, white))
    w.add(xz_rect(0, 555, 0, 555, 555, white))
    w.add(xy_rect(0, 555, 0, 555, 555, white))
    w.add(translate(rotate_y(box((0, 0, 0), (165, 330, 165), white), 15), (265, 0, 295)))
    w.add(translate(rotate_y(box((0, 0, 0), (165, 165, 165), white), -18), (130, 0, 65)))
    w.add(translate(rotate_y(sphere((82.5, 215, 82.5), 50.0, dielectric(1.5)), -18), (130, 0, 65)))
    return w


def cornell_camera(aspect_ratio):
    return camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40, aspect_ratio, 0.0, 10.0)


def _philox_uniforms(n, seed):
    """Host-side scene authoring stream: numpy's Philox, so a scene is identical on every box."""
    return np.random.Generator(np.random.Philox(seed)).random(n)


def procedural_image(width, height, seed=7):
    """A deterministic RGB8 test image (no file or decoder involved)."""
    y, x = np.mgrid[0:height, 0:width]
    r = (x * 255 // max(width - 1, 1)).astype(np.uint8)
    g = (y * 255 // max(height - 1, 1)).astype(np.uint8)
    b = (((x // 8) + (y // 8)) % 2 * 200 + (seed % 50)).astype(np.uint8)
    return np.stack([r, g, b], axis=-1)


_SKYBOX_FILES = dict(earth="textures/earthmap.jpg", mars="textures/mars.jpg", sunset="textures/sunset.jpg",
                     back="textures/skybox/back.jpg", bottom="textures/skybox/bottom.jpg", left="textures/skybox/left.jpg",
                     front="textures/skybox/front.jpg", top="textures/skybox/top.jpg", right="textures/skybox/right.jpg")


def skybox_textures(texture_dir=None):
    """The nine textures of cuda/main.cu:494-503, loaded from `texture_dir` (the directory that
    holds the reference's `textures/`), or deterministic stand-ins of small size when it is None."""
    out = {}
    for k, (name, rel) in enumerate(_SKYBOX_FILES.items()):
        if texture_dir is not None:
            out[name] = load_texture(os.path.join(texture_dir, rel))
        else:
            img = procedural_image(96, 48, seed=7 + 5 * k)
            out[name] = image_texture(96, 48, img)
    return out


def skybox_scene(textures=None, num_small_balls=125, num_boxes=100, seed=1984):
    """The world of the reference's GPU driver (create_world, cuda/main.cu:39-189): three large
    and three small spheres (glass, metal, image-textured), a cluster of `num_boxes` r = 5 spheres
    that the reference wraps in a bvh_node, a two-coloured box of six rects, `num_small_balls`
    random spheres on the floor plane y = 248, the light, and a 555-unit cube of image-textured
    emitters (lambertian_bg) as sky.  The reference rolls its random numbers with curand (seed
    1984, cuda/main.cu:514); here they come from numpy's Philox, drawn in the same order, so the
    layout is of the same kind, not the same numbers."""
    t = textures if textures is not None else skybox_textures()
    rnd = iter(_philox_uniforms(2 + 3 * num_boxes + 10 * num_small_balls, seed))
    rng = lambda lo, hi: lo + (hi - lo) * next(rnd)                       # RND_IN_RANGE
    light = diffuse_light((7, 7, 7))
    white, red, green = lambertian((.73, .73, .73)), lambertian((.65, .05, .05)), lambertian((.12, .45, .15))
    w = sceneobject_list()
    main_x, main_z = rng(-4.0, 4.0), rng(-4.0, 4.0)
    w.add(sphere((200, 278, 400), 30.0, dielectric(1.5)))
    w.add(sphere((100, 278, 350), 30.0, metal((0.7, 0.6, 0.5), 0.0)))
    w.add(sphere((250, 278, 500), 30.0, lambertian(t["earth"])))
    w.add(sphere((230, 263, 300), 15.0, lambertian(t["mars"])))
    w.add(sphere((300, 263, 400), 15.0, metal((0.96, 0.25, 0.25), 0.15)))
    w.add(sphere((100, 315, 405), 15.0, lambertian(t["sunset"])))
    cluster = sceneobject_list()
    for _ in range(num_boxes):
        cluster.add(sphere((rng(200, 250), rng(310, 360), rng(400, 450)), 5.0, white))
    w.add(bvh_node(cluster))
    w.add(box((90, 248, 395), (110, 300, 415), green, red))
    for _ in range(num_small_balls):
        choose, size = next(rnd), rng(5.0, 10.0)
        x, z = rng(50, 400) + main_x + 1.0, rng(250, 550) + main_z + 1.0
        u = [next(rnd) for _ in range(6)]
        if choose < 0.8:
            m = lambertian((u[0] * u[1], u[2] * u[3], u[4] * u[5]))
        elif choose < 0.95:
            m = metal((0.5 * (1 + u[0]), 0.5 * (1 + u[1]), 0.5 * (1 + u[2])), 0.5 * u[3])
        else:
            m = dielectric(1.5)
        w.add(sphere((x, 248 + size, z), size, m))
    d = 555
    w.add(yz_rect(0, d, 0, d, d, lambertian_bg(t["left"])))
    w.add(yz_rect(0, d, 0, d, 0, lambertian_bg(t["right"])))
    w.add(xz_rect(75, 270, 76, 280, 554, light))
    w.add(xz_rect(0, d, 0, d, 0, lambertian_bg(t["bottom"])))
    w.add(xz_rect(0, d, 0, d, d, lambertian_bg(t["top"])))
    w.add(xy_rect(0, d, 0, d, d, lambertian_bg(t["front"])))
    w.add(xy_rect(0, d, 0, d, 0, lambertian_bg(t["back"])))
    return w


def skybox_camera(aspect_ratio):
    """cuda/main.cu:182-187."""
    return camera((450, 278, 200), (0, 278, 555), (0, 1, 0), 40.0, aspect_ratio, 0.0, 10.0)


def stress_scene(n_spheres=1_000_000, seed=42, image=None):
    """BVH stress scene: centres uniform in [-500,500] x [0,50] x [-500,500], radii U(0.1,0.5),
    materials 80/15/5 % lambertian/metal/glass as main.cc:53-70; every tenth lambertian is
    textured (alternating the sin-product checker of cuda/texture.h:44 and an image)."""
    u = _philox_uniforms(n_spheres * 9, seed).reshape(n_spheres, 9)
    if image is None:
        image = procedural_image(256, 128)
    img = image_texture(image.shape[1], image.shape[0], image)
    chk = checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    chk_m, img_m, glass = lambertian(chk), lambertian(img), dielectric(1.5)
    w = sceneobject_list()
    n_lamb = 0
    for k in range(n_spheres):
        c = (-500 + 1000 * u[k, 0], 50 * u[k, 1], -500 + 1000 * u[k, 2])
        r = 0.1 + 0.4 * u[k, 3]
        choose = u[k, 4]
        if choose < 0.8:
            n_lamb += 1
            if n_lamb % 10 == 0:
                m = chk_m if (n_lamb // 10) % 2 else img_m
            else:
                m = lambertian((u[k, 5] * u[k, 6], u[k, 6] * u[k, 7], u[k, 7] * u[k, 8]))
        elif choose < 0.95:
            m = metal((0.5 + 0.5 * u[k, 5], 0.5 + 0.5 * u[k, 6], 0.5 + 0.5 * u[k, 7]), 0.5 * u[k, 8])
        else:
            m = glass
        w.add(sphere(c, r, m))
    return w


def stress_flat(n_spheres=1_000_000, seed=42, image=None) -> FlatScene:
    """stress_scene() flattened directly with numpy (a million Python objects are slow to make):
    the same spheres, materials and textures, in the same order."""
    u = _philox_uniforms(n_spheres * 9, seed).reshape(n_spheres, 9)
    if image is None:
        image = procedural_image(256, 128)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    geom = np.zeros((n_spheres, 6))
    geom[:, 0] = -500 + 1000 * u[:, 0]
    geom[:, 1] = 50 * u[:, 1]
    geom[:, 2] = -500 + 1000 * u[:, 2]
    geom[:, 3] = 0.1 + 0.4 * u[:, 3]
    choose = u[:, 4]
    is_l, is_m = choose < 0.8, (choose >= 0.8) & (choose < 0.95)
    n_lamb = np.cumsum(is_l)
    textured = is_l & (n_lamb % 10 == 0)
    use_chk = textured & ((n_lamb // 10) % 2 == 1)
    use_img = textured & ~use_chk
    # materials 0,1,2 = checker lambertian, image lambertian, glass; then one per plain sphere
    plain = ~(textured | ~(is_l | is_m))
    prim_mat = np.zeros(n_spheres, np.int32)
    prim_mat[use_chk], prim_mat[use_img], prim_mat[~(is_l | is_m)] = 0, 1, 2
    prim_mat[plain] = 3 + np.arange(int(plain.sum()))
    pl, pm = plain & is_l, plain & is_m
    mat_type = np.concatenate([[RT_LAMBERTIAN, RT_LAMBERTIAN, RT_DIELECTRIC],
                               np.where(is_l[plain], RT_LAMBERTIAN, RT_METAL)]).astype(np.int32)
    param = np.zeros((3 + int(plain.sum()), 5))
    param[2] = (1, 1, 1, 0, 1.5)
    lam = np.stack([u[:, 5] * u[:, 6], u[:, 6] * u[:, 7], u[:, 7] * u[:, 8], np.zeros(n_spheres), np.zeros(n_spheres)], 1)
    met = np.stack([0.5 + 0.5 * u[:, 5], 0.5 + 0.5 * u[:, 6], 0.5 + 0.5 * u[:, 7], 0.5 * u[:, 8], np.zeros(n_spheres)], 1)
    param[3:] = np.where(is_l[plain][:, None], lam[plain], met[plain])
    mat_tex = -np.ones(3 + int(plain.sum()), np.int32)
    mat_tex[0], mat_tex[1] = 2, 3        # textures: 0,1 solid colours of the checker, 2 checker, 3 image
    return FlatScene(
        prim_type=np.zeros(n_spheres, np.int32), prim_geom=geom, prim_mat=prim_mat,
        mat_type=mat_type, mat_param=param, mat_tex=mat_tex,
        tex_type=np.array([RT_TEX_SOLID, RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE], np.int32),
        tex_color=np.array([[0.2, 0.3, 0.1], [0.9, 0.9, 0.9], [0, 0, 0], [0, 0, 0]], np.float64),
        tex_a=np.array([0, 0, 0, image.shape[1]], np.int32), tex_b=np.array([0, 0, 1, image.shape[0]], np.int32),
        tex_off=np.zeros(4, np.int64), texels=image.reshape(-1), background=RT_BG_SKY, bg_color=(0.0, 0.0, 0.0))


def stress_camera(aspect_ratio):
    return camera((0, 60, -120), (0, 10, 0), (0, 1, 0), 40, aspect_ratio, 0.0, 10.0)
