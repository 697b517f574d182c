"""raytrace_amd — MI355X (gfx950) render path behind the C-ABI of include/rt.h.

`scene` holds the reference's scene vocabulary (sphere, rects, materials, textures, camera,
sceneobject_list) and flattens it; `api` binds librt_amd.so through ctypes.  There is no CPU
fallback: importing `api` without the built library raises.
"""
from . import scene  # noqa: F401

__all__ = ["scene", "api", "dist", "ppm"]
