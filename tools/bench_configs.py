"""Measures the benchmark configurations other than the headline one on one GPU:
C3 Cornell 1080p, C4 BVH stress (N spheres, textures) at 4K, C5 camera-path frames."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S

def timed(ctx, params, buf, reps=2):
    best = None
    for _ in range(reps):
        ctx.render(params, buf.data_ptr())
        st = ctx.stats()
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    return best

out = {}
P = api.Context.params
ctx = api.Context(0)
aspect = float(np.float32(16 / 9))

# C3: Cornell box 1920x1080; 1000 spp would be 2.07 G samples: time 100 spp and scale
t0 = time.time()
ctx.upload_scene(S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0)))
ctx.set_camera(**S.cornell_camera(aspect).desc())
buf = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
spp = int(os.environ.get("C3_SPP", "100"))
st = timed(ctx, P(1920, 1080, spp, 50, 1), buf)
out["C3_cornell_1080p"] = dict(spp=spp, ms=st["kernel_ms"], mrays_s=st["rays"] / st["kernel_ms"] / 1e3, rays=st["rays"],
                               rays_per_sample=st["rays"] / st["samples"], ms_at_1000spp=st["kernel_ms"] * 1000 / spp)
print(json.dumps(out["C3_cornell_1080p"]), flush=True)

# C4: BVH stress
n = int(os.environ.get("C4_SPHERES", "1000000"))
t0 = time.time(); flat = S.stress_flat(n, seed=42); t_scene = time.time() - t0
t0 = time.time(); ctx.upload_scene(flat, accel=api.RT_ACCEL_BVH); t_up = time.time() - t0
ctx.set_camera(**S.stress_camera(aspect).desc())
buf = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
spp = int(os.environ.get("C4_SPP", "64"))
st = timed(ctx, P(3840, 2160, spp, 50, 1), buf)
out["C4_stress_4k"] = dict(spheres=n, spp=spp, ms=st["kernel_ms"], mrays_s=st["rays"] / st["kernel_ms"] / 1e3, rays=st["rays"],
                           rays_per_sample=st["rays"] / st["samples"], bvh_nodes=st["bvh_nodes"], bvh_depth=st["bvh_depth"],
                           scene_build_s=t_scene, upload_and_bvh_build_s=t_up)
print(json.dumps(out["C4_stress_4k"]), flush=True)
cst = timed(ctx, P(3840, 2160, 4, 50, 1, flags=api.RT_FLAG_COUNTERS), buf, reps=1)
print("C4 per ray:", {k: round(cst[k] / cst["rays"], 2) for k in ("box_tests", "sphere_tests", "sphere_hits")},
      "trav lane util", cst["phase_clocks"][0] / (64.0 * cst["phase_clocks"][1]), flush=True)

# C5: camera path, 1080p frames of the RTIOW scene, yaw +-0.125 deg/frame (cuda/main.cu:227-238)
ctx.upload_scene(S.flatten(S.rtiow_scene()))
cam = S.rtiow_camera(aspect)
frames = int(os.environ.get("C5_FRAMES", "8")); spp = int(os.environ.get("C5_SPP", "256"))
# the driver of raytrace_amd/sequence.py with a second context on a second stream: a frame is
# launched before the one before it is read back (out_dir=None: no files)
from raytrace_amd import sequence
other = api.Context(0)
side = torch.cuda.Stream()
other.set_stream(side.cuda_stream)
other.upload_scene(S.flatten(S.rtiow_scene()))
torch.cuda.synchronize(); t0 = time.time()
done = sequence.render_sequence(ctx, cam, frames, 1920, 1080, spp, None, max_depth=50, seed=1000, overlap_ctx=other)
rays = sum(d[1] for d in done)
dt = time.time() - t0
out["C5_camera_path"] = dict(frames=frames, spp=spp, s_per_frame=dt / frames, mrays_s=rays / dt / 1e6,
                             s_for_240_frames=dt / frames * 240)
print(json.dumps(out["C5_camera_path"]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
