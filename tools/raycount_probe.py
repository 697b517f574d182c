"""Which objects of the skybox world make the GPU trace more rays than the double-precision checker?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytrace_amd import api, scene as S
from oracle.pyoracle import Oracle
o = Oracle("f64")
full = S.skybox_scene()
objs = full.objects
groups = {"all": objs, "big6": objs[:6] + objs[-7:], "cluster": objs[6:7] + objs[-7:], "box": objs[7:8] + objs[-7:],
          "balls": objs[8:-7] + objs[-7:], "walls": objs[-7:]}
cam = S.skybox_camera(1.5).desc()
cd = o.camera(**cam)
ctx = api.Context(0)
for name, g in groups.items():
    flat = S.flatten(S.sceneobject_list(list(g)), background=S.RT_BG_SOLID)
    ctx.upload_scene(flat); ctx.set_camera(**cam)
    ctx.render_host(api.Context.params(96, 64, 32, 50, 27)); r = ctx.stats()["rays"]
    ctx.render_host(api.Context.params(96, 64, 32, 50, 27, flags=api.RT_FLAG_NO_REFINE)); r2 = ctx.stats()["rays"]
    _, want = o.render(flat, cd, 96, 64, 32, 50, 27, threads=8)
    print(f"{name:8s} prims {flat.n_prims:4d} gpu {r} noref {r2} oracle {want} diff {100*(r-want)/want:+.2f}%", flush=True)
# the same balls world moved so that coordinates are small
for shift in ((0, 0, 0), (-278, -278, -278), (5000, 0, 0)):
    g = groups["balls"]
    flat = S.flatten(S.translate(S.sceneobject_list(list(g)), shift), background=S.RT_BG_SOLID)
    c2 = dict(cam); c2["lookfrom"] = tuple(np.add(cam["lookfrom"], shift)); c2["lookat"] = tuple(np.add(cam["lookat"], shift))
    ctx.upload_scene(flat); ctx.set_camera(**c2)
    ctx.render_host(api.Context.params(96, 64, 32, 50, 27)); r = ctx.stats()["rays"]
    _, want = o.render(flat, o.camera(**c2), 96, 64, 32, 50, 27, threads=8)
    print(f"shift {shift} gpu {r} oracle {want} diff {100*(r-want)/want:+.2f}%", flush=True)
