#!/usr/bin/env python3
"""Benchmark of the render path on N MI355X of one node: Mrays/s and ms per frame.

    python bench.py                          headline: --config c2 on one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

--config   workload (a step is one frame)                                        default partition
  c2       RTIOW random spheres (the reference's default world, 487 spheres),     tiles
           1920x1080 @100 spp, depth 50   [the configuration the metric is quoted on]
  c3       Cornell box (rects + diffuse_light), 1920x1080 @1000 spp                tiles
  c4       BVH stress: 10^6 spheres, checker and image textures, 3840x2160 @64     tiles
  c5       camera path (cuda/main.cu:220-279) over the RTIOW scene,                frames+samples
           1920x1080 @256 spp, a different camera and seed every step

--partition (N > 1)
  tiles            rank g renders the interleaved 4-row tiles g, g+N, ...; the partial framebuffers
                   are gathered over RCCL inside the timed region, step k's on rank k % N (the receive
                   side of the gather and the row index are shared out like the rendering;
                   --gather-root 0: always on rank 0)
  samples          rank g renders the whole frame for its share of the samples as integer sums
                   (RT_FLAG_FIXED_SUMS); an int64 RCCL reduce (to rank k % N for step k) and
                   rt_resolve_sums give the frame, bit for bit the single-GPU one
  frames+samples   (c5) frames are dealt to the ranks in rounds of N, no exchange; the frames of
                   a last, incomplete round are sample-split over all ranks as above

Three frames are in flight: frames k+1 and k+2 are launched (streams, contexts and buffers of their
own) before frame k is collected, so the timed K steps are K complete frames at the throughput a frame
sequence gets; the time of one launch alone, start to end, is reported beside it
(roofline.kernel_ms).  A ray is one closest-hit query (main.cc:18), counted on the device.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE_ROWS = 4
PEAK_FP32_TFLOPS = 157.3          # MI355X FP32 vector peak (256 CU x 128 lanes x 2 x 2.4 GHz)

# algorithmic flops per unit of work, fixed constants: sphere test 25 (+24 when a root is
# accepted), box test 24, rect test 6, scatter ~47 (80/15/5 % lambertian/metal/glass mix of
# 45/50/70), miss (sky) 20, camera ray 40
F_SPHERE, F_SPHERE_HIT, F_BOX, F_RECT, F_SHADE, F_MISS, F_CAMERA = 25, 24, 24, 6, 47, 20, 40


def executed_flops(st):
    misses = st["rays"] - st["shades"]
    return ((st["sphere_tests"] + st["wide_tests"]) * F_SPHERE + st["sphere_hits"] * F_SPHERE_HIT +
            st["box_tests"] * F_BOX + st["rect_tests"] * F_RECT + st["shades"] * F_SHADE +
            misses * F_MISS + st["samples"] * F_CAMERA)


NODE_BYTES = 32                   # one node visit of a walk out of HBM reads a 32-byte quantised node (two box tests)
PEAK_HBM_GBS = 8000.0             # MI355X HBM3E


def _traffic_file(config):
    import glob
    name = "hbm_traffic.json" if config == "c2" else f"{config}_hbm_traffic.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except (OSError, ValueError):
        return None


def measured_traffic(config):
    """HBM bytes per render launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE are
    collected with rocprofv3 in runs of their own, so they cannot be read inside this process)."""
    d = _traffic_file(config)
    return d.get("hbm_bytes_per_launch") if d else None


def memory_side(config, cst, kernel_ms, width, height):
    """The bandwidth side of the roofline, for a reader to check `bound` against: the bytes the
    algorithm asks the memory system for per launch (framebuffer + one 32-byte node per node
    visit when the tree is walked out of HBM; counting launch of the same kernel), what of that
    missed L2 (committed PMC passes of the same command), and both over the launch's duration."""
    fb = width * height * 16
    walked_from_hbm = config == "c4"
    nodes = (cst["box_tests"] // 2) * NODE_BYTES if walked_from_hbm else 0
    d = _traffic_file(config) or {}
    miss = d.get("hbm_bytes_per_launch")
    wr = d.get("WRITE_SIZE_KB_per_launch")
    out = {"framebuffer_bytes": fb, "node_bytes_requested": nodes, "algorithmic_bytes_per_launch": fb + nodes,
           "requested_GBs": (fb + nodes) / (kernel_ms * 1e-3) / 1e9,
           "l2_miss_bytes": miss, "l2_miss_GBs": miss / (kernel_ms * 1e-3) / 1e9 if miss else None,
           "l2_miss_frac_of_hbm_peak": miss / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if miss else None,
           "write_bytes": wr * 1024 if wr else None, "peak_hbm_GBs": PEAK_HBM_GBS,
           "note": "node_bytes_requested = 32 B x node visits, served by L1 / L2 / Infinity Cache (the scene stays resident); "
                   "l2_miss_bytes = FETCH_SIZE x2 + WRITE_SIZE of the committed PMC passes (upper bound)"}
    return out


def lone_frames(name, device, stream_ptr, frames=3):
    """One more configuration under this run's clock: `frames` launches of its frame, one at a time
    (HIP events around each, on the stream it runs on), then the counting launch of the same kernel
    for the executed work.  The frame is the configuration's own size and sample count."""
    import torch
    from raytrace_amd import api
    wl = Workload(name)
    ctx = api.Context(device)
    ctx.set_stream(stream_ptr)
    t0 = time.perf_counter()
    ctx.upload_scene(wl.flat, accel=wl.accel)
    upload_s = time.perf_counter() - t0
    cam, seed = wl.camera_of_step(1 if wl.path else 0)
    ctx.set_camera(**cam)
    buf = torch.zeros((wl.H, wl.W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ms, st = [], None
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.render(api.Context.params(wl.W, wl.H, wl.spp, wl.depth, seed), buf.data_ptr())
        st = ctx.stats()                                     # waits for the launch
        ms.append(st["kernel_ms"])
    wall_ms = (time.perf_counter() - t0) / frames * 1e3
    ctx.render(api.Context.params(wl.W, wl.H, wl.spp, wl.depth, seed, flags=api.RT_FLAG_COUNTERS), buf.data_ptr())
    cst = ctx.stats()
    ctx.close()
    del buf
    kernel_ms = sum(ms) / len(ms)
    flops = executed_flops(cst)
    out = {"workload": wl.label, "width": wl.W, "height": wl.H, "spp": wl.spp, "frames": frames,
           "kernel_ms": kernel_ms, "kernel_ms_each": ms, "wall_ms_per_frame": wall_ms,
           "rays_per_frame": st["rays"], "mrays_s": st["rays"] / kernel_ms / 1e3,
           "frac": flops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "achieved_tflops": flops / (kernel_ms * 1e-3) / 1e12,
           "flops_per_ray": flops / max(cst["rays"], 1),
           "tests_per_ray": {k: cst[k] / max(cst["rays"], 1) for k in ("box_tests", "sphere_tests", "sphere_hits", "rect_tests", "wide_tests")},
           "traversal_lane_utilisation": cst["phase_clocks"][0] / (64.0 * cst["phase_clocks"][1]) if cst["phase_clocks"][1] else None,
           "lanes_per_pixel": st["lanes_per_pixel"], "bvh_depth": st["bvh_depth"], "upload_s": upload_s}
    if name == "c4":
        out["memory_side"] = memory_side(name, cst, kernel_ms, wl.W, wl.H)
    return out


def reconstruction_stage(device, stream_ptr):
    """The low-to-high sample-count reconstruction stage on a frame of the camera path (1920x1080): guides +
    a-trous passes on the 16-sample frame, against the 256-sample frame of the same camera.  Bandwidth-bound;
    parity unpinned (the reference holds no code or output of such a stage), so the figure of merit is PSNR."""
    import torch
    from raytrace_amd import api
    wl = Workload("c5")
    W, H, low_spp, high_spp, passes = wl.W, wl.H, 16, wl.spp, 4
    ctx = api.Context(device)
    ctx.set_stream(stream_ptr)
    ctx.upload_scene(wl.flat, accel=wl.accel)
    cam, seed = wl.camera_of_step(1)
    ctx.set_camera(**cam)
    low, high, out = (torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(3))
    guides = torch.zeros((2, H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render(api.Context.params(W, H, high_spp, wl.depth, seed), high.data_ptr())
    ctx.render(api.Context.params(W, H, low_spp, wl.depth, seed), low.data_ptr())
    low_ms = ctx.stats()["kernel_ms"]
    g_ms, r_ms = [], []
    for _ in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        ctx.render_guides(W, H, guides.data_ptr())
        ev[1].record()
        ctx.reconstruct(low.data_ptr(), guides.data_ptr(), W, H, out.data_ptr(), iterations=passes)
        ev[2].record()
        torch.cuda.synchronize()
        g_ms.append(ev[0].elapsed_time(ev[1]))
        r_ms.append(ev[1].elapsed_time(ev[2]))

    def psnr(a, spp):
        shown = lambda x, n: torch.sqrt(torch.clamp(x[..., :3] / n, 0.0, 1.0))
        return float(-10.0 * torch.log10(torch.mean((shown(a, spp) - shown(high, high_spp)) ** 2)))

    # the upscaler half: the same view at half the resolution and 4 samples, reconstructed, then brought to
    # 1920x1080 by rt_upscale; plain bilinear interpolation of the same low frame beside it
    lw, lh, up_spp = W // 2, H // 2, 4
    small, small_den = (torch.zeros((lh, lw, 4), dtype=torch.float32, device="cuda") for _ in range(2))
    g_small = torch.zeros((2, lh, lw, 4), dtype=torch.float32, device="cuda")
    up = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render(api.Context.params(lw, lh, up_spp, wl.depth, seed), small.data_ptr())
    small_ms = ctx.stats()["kernel_ms"]
    u_ms = []
    for _ in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        ctx.render_guides(lw, lh, g_small.data_ptr())
        ctx.reconstruct(small.data_ptr(), g_small.data_ptr(), lw, lh, small_den.data_ptr(), iterations=passes)
        ctx.upscale(small_den.data_ptr(), g_small.data_ptr(), lw, lh, guides.data_ptr(), W, H, up.data_ptr())
        ev[1].record()
        torch.cuda.synchronize()
        u_ms.append(ev[0].elapsed_time(ev[1]))
    bil = torch.nn.functional.interpolate((small[..., :3] / up_spp).permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0) * up_spp
    upscaling = {"workload": f"{lw}x{lh} @{up_spp} reconstructed and brought to {W}x{H}, against the {W}x{H} @{high_spp} frame",
                 "low_frame_ms": small_ms, "guides_reconstruct_upscale_ms": min(u_ms),
                 "psnr_bilinear_db": psnr(bil, up_spp), "psnr_upscaled_db": psnr(up, up_spp)}
    ctx.close()
    g, r = min(g_ms), min(r_ms)
    algo = W * H * (48 + 16 + (passes - 1) * (36 + 16))       # first pass: frame, normal + distance, albedo + object in, colour out; then per pass: colour, normal + distance, object in, colour out
    return {"workload": f"{W}x{H}, frame 1 of the camera path: {low_spp}-sample frame reconstructed towards the {high_spp}-sample frame",
            "low_frame_ms": low_ms, "guides_ms": g, "reconstruct_ms": r, "passes": passes,
            "psnr_low_db": psnr(low, low_spp), "psnr_reconstructed_db": psnr(out, low_spp),
            "roofline": {"bound": "hbm", "achieved": algo / (r * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": algo / (r * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes": algo},
            "upscaling": upscaling,
            "parity": "unpinned: the reference names the stage (README.md:7) and holds no code, weights or output of it"}


class Workload:
    """Scene, camera and frame size of one --config."""

    def __init__(self, name, spp=None):
        import numpy as np
        from raytrace_amd import api, scene as S
        self.name = name
        self.depth, self.seed = 50, 1
        self.accel = api.RT_ACCEL_BVH
        self.path = False                       # the camera moves from step to step
        if name in ("c1", "c2", "c5"):
            self.W, self.H, self.spp = (1200, 800, 500) if name == "c1" else (1920, 1080, 100 if name == "c2" else 256)
            self.flat = S.flatten(S.rtiow_scene())
            self.camera = S.rtiow_camera(float(np.float32(self.W / self.H)))
            self.n_spheres, self.n_rects = 487, 0
            self.path = name == "c5"
            self.label = ("configs[0] on the GPU: the frame main.cc renders by default, 1200x800 @500spp depth 50 (main.cc:90-106)" if name == "c1" else
                          "configs[1]: RTIOW random-spheres scene (the reference's default world, 487 spheres, "
                          "g++ build) 1920x1080 @100spp depth 50" if name == "c2" else
                          "configs[4]: camera-path sequence (cuda/main.cu:220-279) over the RTIOW scene, "
                          "1920x1080 @256spp depth 50, one frame per step")
        elif name == "c3":
            self.W, self.H, self.spp = 1920, 1080, 1000
            self.flat = S.flatten(S.cornell_scene(), background=S.RT_BG_SOLID, bg_color=(0, 0, 0))
            self.camera = S.cornell_camera(float(np.float32(self.W / self.H)))
            self.n_spheres, self.n_rects = 2, 6
            self.accel = api.RT_ACCEL_AUTO
            self.label = ("configs[2]: Cornell box (5 walls + diffuse_light rect, glass and metal r=30 spheres) "
                          "1920x1080 @1000spp depth 50")
        elif name == "c4":
            self.W, self.H, self.spp = 3840, 2160, 64
            self.flat = S.stress_flat(1_000_000, seed=42)
            self.camera = S.stress_camera(float(np.float32(self.W / self.H)))
            self.n_spheres, self.n_rects = 1_000_000, 0
            self.label = ("configs[3]: BVH stress scene, 10^6 random spheres with checker and image textures, "
                          "3840x2160 @64spp depth 50 (host-built binned-SAH tree)")
        else:
            raise SystemExit(f"unknown config {name}")
        if spp:
            self.spp = spp
            self.label += f" [--spp {spp}]"

    def bruteforce_flops_per_ray(self):
        """What the reference's linear scan (sceneobject_list.h:26-39) spends on a ray."""
        return self.n_spheres * F_SPHERE + self.n_rects * F_RECT + 70

    def camera_of_step(self, k):
        """c5: the camera of frame k (frames follow each other; steps are dealt in order)."""
        from raytrace_amd import scene as S
        if not self.path:
            return self.camera.desc(), self.seed
        if not hasattr(self, "_walk"):
            self._walk, self._walk_frame = S.rtiow_camera(self.camera.aspect_ratio), -1
        if k < self._walk_frame:                 # an earlier frame again: walk from the start
            self._walk, self._walk_frame = S.rtiow_camera(self.camera.aspect_ratio), -1
        while self._walk_frame < k:
            self._walk_frame += 1
            S.update_scene(self._walk, self._walk_frame)
        return self._walk.desc(), self.seed + k


def cpu_baseline(wl):
    """The CPU side of the same workload on the host cores of this box, on a bounded sample.
    c2 / c5 (spheres only): the reference's own main.cc (compiled into oracle/_ref), one core,
    glibc rand() like the original binary.  c3 / c4 use features main.cc lacks: the C restatement
    (linear scan, double precision) on the job's host threads."""
    from oracle.pyoracle import Oracle, Reference
    from raytrace_amd import scene as S
    W, H = wl.W, wl.H
    cam = wl.camera.desc()
    if wl.name in ("c2", "c5") and Reference.available():
        ref = Reference()
        ref.load_scene(S.rtiow_records())
        ref.set_camera(cam["lookfrom"], cam["lookat"], cam["vup"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus_dist"])
        t0 = time.perf_counter()
        _, rays = ref.render_libc(W, H, 16, wl.depth, step=(3, 3))
        dt = time.perf_counter() - t0
        return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": f"{W}x{H} frame, every 3rd pixel of every 3rd row, 16 spp, depth {wl.depth}, 1 thread",
                "seconds": dt, "rays": rays}
    orc = Oracle("f64")
    threads = 1 if wl.name in ("c2", "c5") else max(1, min(16, os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    step, spp = {"c2": ((3, 3), 16), "c5": ((3, 3), 16), "c3": ((1, 1), 256), "c4": ((20, 20), 1)}[wl.name]
    cd = orc.camera(cam["lookfrom"], cam["lookat"], cam["vup"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus_dist"])
    t0 = time.perf_counter()
    _, rays = orc.render(wl.flat, cd, W, H, spp, wl.depth, wl.seed, step=step, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{W}x{H} frame, " + ("every pixel" if step == (1, 1) else f"every {step[0]}th pixel of every {step[1]}th row") +
                      f", {spp} spp, depth {wl.depth}, "
                      f"{threads} thread(s), linear scan over {wl.flat.n_prims} primitives in double",
            "seconds": dt, "rays": rays}


def cpu_baseline_all_cores(wl):
    """c2: the same sample on the host cores this job may use (16 per GPU on the test boxes), with the
    C restatement: the reference itself cannot run on several threads (utils.h:25 draws from the
    process-wide rand()), the restatement draws from keyed streams and splits rows over OpenMP
    threads."""
    from oracle.pyoracle import Oracle
    threads = max(1, min(16, os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    orc = Oracle("f64")
    cam = wl.camera.desc()
    cd = orc.camera(cam["lookfrom"], cam["lookat"], cam["vup"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus_dist"])
    t0 = time.perf_counter()
    _, rays = orc.render(wl.flat, cd, wl.W, wl.H, 16, wl.depth, wl.seed, step=(2, 2), threads=threads)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{wl.W}x{wl.H} frame, every 2nd pixel of every 2nd row, 16 spp, depth {wl.depth}, {threads} threads",
            "seconds": dt, "rays": rays}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 100 frames of c2, about 1.7 s on one GPU, so that a sampler outside the process "
                         "sees the timed region; 10 of the other configurations)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--partition", choices=["tiles", "samples", "frames+samples"], default=None,
                    help="how N > 1 ranks share the work (default: tiles; c5: frames+samples)")
    ap.add_argument("--spp", type=int, default=None, help="override the config's samples per pixel (rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="c2 on one GPU: do not render the lone frames of c3, c4 and c5 after the timed region")
    ap.add_argument("--accel", choices=["bvh", "list"], default="bvh")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearse the N-rank path on fewer GPUs (ranks share devices, shards exchanged via host)")
    ap.add_argument("--frames-in-flight", type=int, choices=[1, 2, 3], default=3,
                    help="2 / 3: frames k+1 (and k+2) are launched on streams of their own before frame k is collected "
                         "(3 measured 0.6 %% faster than 2 on a whole frame, 2.3 %% on an eighth of one); "
                         "1: one frame at a time (what the kernel-trace profile is taken with)")
    ap.add_argument("--float-nodes", action="store_true",
                    help="walk a tree that lives in HBM through its 64-byte float nodes, not the 32-byte quantised ones (A/B)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with one rank: set the process group up all the same and send every frame through the partition's "
                         "collectives (a gather or an integer reduce among one rank), so that the RCCL code path runs on a one-GPU box")
    ap.add_argument("--gather-root", choices=["rotate", "0"], default="rotate",
                    help="which rank assembles step k's frame (tiles, samples): k %% N, so that the gather's receive side, the row "
                         "index and the resolve are shared out like the rendering, or always rank 0")
    ap.add_argument("--check", action="store_true",
                    help="rank 0 also renders every timed frame alone and compares it bit for bit with the assembled one")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100 if args.config == "c2" else 10

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytrace_amd import api, dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    partition = args.partition or ("frames+samples" if args.config == "c5" else "tiles")
    if partition == "frames+samples" and args.config != "c5":
        raise SystemExit("--partition frames+samples is the camera-path config's (c5)")
    device = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    comm_dev = "cuda" if args.backend == "nccl" else "cpu"
    dist_on = world > 1 or args.rehearse_collectives
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                       # not launched by torch.distributed.run
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
        # the first collectives on a communicator set up its peer connections: done here, untimed,
        # whatever --warmup is
        probe = torch.zeros(16, device=comm_dev)
        probe_i = torch.zeros(16, dtype=torch.int64, device=comm_dev)
        for root in range(world if args.gather_root == "rotate" else 1):
            dist.gather(probe, [torch.empty_like(probe) for _ in range(world)] if rank == root else None, dst=root)
            dist.reduce(probe_i, dst=root)
        dist.barrier()

    wl = Workload(args.config, args.spp)
    W, H, SPP, DEPTH = wl.W, wl.H, wl.spp, wl.depth
    accel = api.RT_ACCEL_LIST if args.accel == "list" else wl.accel

    # Frames in flight: frame k+1 (and k+2) is launched on a stream of its own (its own context,
    # counters and framebuffer) before frame k is collected, so it starts in the tail of frame k,
    # where the last long paths leave most of the chip idle, and renders while frame k's shards
    # travel.  The collectives stay on torch's current stream.  There is one framebuffer more than
    # contexts: the gather of frame k finds room on the chip only in the tail of the frame after it
    # (its kernels wait for workgroup slots like any other), and with the spare buffer the next
    # launch does not wait for it.
    n_ctx = args.frames_in_flight
    n_buf = n_ctx + 1 if n_ctx > 1 else 1
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(n_ctx)]
    ctxs = []
    for s in streams:
        c = api.Context(device)
        c.set_stream(s.cuda_stream)
        c.upload_scene(wl.flat, accel=accel)
        c.set_camera(**wl.camera.desc())
        ctxs.append(c)
    ctx = ctxs[0]
    build_ms = ctx.stats()["build_ms"]
    # resolve / quantise of a reduced frame run on torch's stream, behind the reduce (a context
    # without a scene: those two calls need none)
    post = api.Context(device)
    post.set_stream(main_stream.cuda_stream)

    # How step k is shared out.  mode "tiles": this rank's interleaved row tiles, float image,
    # gather.  mode "samples": this rank's sample window of the whole frame, integer sums,
    # reduce + resolve.  mode "own": the whole frame on one rank, nothing exchanged (frame-parallel
    # rounds of c5; and every step of a single-GPU run).
    full_rounds = (args.steps // world) * world if partition == "frames+samples" else 0

    def root_of(k):
        """The rank that ends up with step k's frame when the ranks share it."""
        return k % world if args.gather_root == "rotate" else 0

    def plan(k, n_steps):
        if not dist_on:
            return "own", 0
        if partition == "tiles":
            return "tiles", None
        if partition == "samples":
            return "samples", None
        rounds = (n_steps // world) * world
        return ("own", k % world) if k < rounds else ("samples", None)

    tiles = (TILE_ROWS, rank, world) if dist_on else (0, 0, 0)
    max_rows = rdist.max_shard_rows(H, TILE_ROWS, world) if (dist_on and partition == "tiles") else H
    extra_flags = api.RT_FLAG_FLOAT_NODES if args.float_nodes else 0

    def P(*a, flags=0, **kw):
        return api.Context.params(*a, flags=flags | extra_flags, **kw)
    fbufs = [torch.zeros((max_rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(n_buf)]
    ibufs = [torch.zeros((H, W, 4), dtype=torch.int64, device="cuda") for _ in range(n_buf)] if (dist_on and partition != "tiles") else None
    rgb8 = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if wl.path else None
    my_begin, my_count = rdist.sample_window(SPP, rank, world)
    rendered = [torch.cuda.Event() for _ in range(n_buf)]    # the frame in buffer i is complete
    consumed = [torch.cuda.Event() for _ in range(n_buf)]    # buffer i has been collected and may be overwritten
    torch.cuda.synchronize()
    for i in range(n_buf):
        consumed[i].record(main_stream)

    def launch(k, n_steps, ev=None):
        c, i = k % n_ctx, k % n_buf                          # launches of one context run in stream order
        mode, owner = plan(k, n_steps)
        streams[c].wait_event(consumed[i])
        if ev:
            ev[0].record(streams[c])
        if mode != "own" or owner == rank:
            cam, seed = wl.camera_of_step(k)
            if wl.path:
                ctxs[c].set_camera(**cam)
            if mode == "tiles":
                ctxs[c].render(P(W, H, SPP, DEPTH, seed, tiles=tiles), fbufs[i].data_ptr())
            elif mode == "samples":
                if my_count > 0:
                    ctxs[c].render(P(W, H, my_count, DEPTH, seed, sample_begin=my_begin, flags=api.RT_FLAG_FIXED_SUMS), ibufs[i].data_ptr())
                else:
                    with torch.cuda.stream(streams[c]):
                        ibufs[i].zero_()
            else:
                ctxs[c].render(P(W, H, SPP, DEPTH, seed), fbufs[i].data_ptr())
                if wl.path:                                  # the sequence driver's write_color step (cuda/main.cu:587)
                    ctxs[c].quantize(fbufs[i].data_ptr(), W * H, SPP, rgb8[i].data_ptr())
        if ev:
            ev[1].record(streams[c])
        rendered[i].record(streams[c])

    def collect(k, n_steps):
        """Returns the finished frame where this rank holds it (float image; c5: RGB8), else None."""
        c, i = k % n_ctx, k % n_buf
        mode, owner = plan(k, n_steps)
        main_stream.wait_event(rendered[i])
        frame, root = None, root_of(k)
        if mode == "tiles":
            frame = rdist.gather_frame(fbufs[i].cpu() if args.backend == "gloo" else fbufs[i], H, TILE_ROWS, rank, world, dst=root,
                                       always=args.rehearse_collectives)
        elif mode == "samples":
            if args.backend == "gloo":
                host = ibufs[i].cpu()
                rdist.reduce_sums(host, rank, world, dst=root, always=args.rehearse_collectives)
                if rank == root:
                    ibufs[i].copy_(host)
            else:
                rdist.reduce_sums(ibufs[i], rank, world, dst=root, always=args.rehearse_collectives)
            if rank == root:
                post.resolve_sums(ibufs[i].data_ptr(), W * H, fbufs[i].data_ptr())
                if wl.path:
                    post.quantize(fbufs[i].data_ptr(), W * H, SPP, rgb8[i].data_ptr())
                frame = rgb8[i] if wl.path else fbufs[i][:H]
        elif owner == rank:
            frame = rgb8[i] if wl.path else fbufs[i][:H]
        consumed[i].record(main_stream)
        return frame

    kept = {}                                                # --check: the frames this rank ended up holding

    def run(n, events=None, keep=False):
        frame, launched = None, 0
        for k in range(n):
            while launched < n and launched < k + n_buf:     # up to n_buf - 1 frames ahead of the collect
                launch(launched, n, events[launched] if events else None)
                launched += 1
            frame = collect(k, n)
            if keep and frame is not None:
                kept[k] = frame.clone()
        return frame

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    frame = run(args.steps, events)
    fence()
    dt = time.perf_counter() - t0

    if args.check:                                           # the same steps again, untimed, kept
        run(args.steps, keep=True)
        torch.cuda.synchronize()

    # rays of the timed region, whole job: every launch's device counter
    # (a context's counter holds its last launch; all frames of a fixed camera are equal)
    def rays_of_step(k, n_steps):
        """Device ray count of this rank's launch for step k (re-rendered alone, untimed)."""
        mode, owner = plan(k, n_steps)
        if mode == "own" and owner != rank:
            return 0
        launch(k, n_steps)
        torch.cuda.synchronize()
        return ctxs[k % n_ctx].stats()["rays"]

    if wl.path:
        my_rays = sum(rays_of_step(k, args.steps) for k in range(args.steps))
    else:
        my_rays = rays_of_step(0, args.steps) * args.steps
    for i in range(n_buf):
        consumed[i].record(main_stream)

    # the duration of one launch, start to end.  A launch queued behind the frame before it waits
    # for that frame's waves to leave the chip, and an event pair around it times the wait with
    # it: with two frames in flight it is taken from launches that run alone, right after the
    # timed region (same kernel, same arguments, same stream).
    lone_step = 0 if plan(0, args.steps)[0] != "own" or plan(0, args.steps)[1] == rank else rank
    if args.frames_in_flight >= 2 and args.steps > 1:
        lone = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for _ in range(min(args.steps, 5))]
        for ev in lone:
            launch(lone_step, args.steps, ev)
            torch.cuda.synchronize()
        kernel_ms = sum(a.elapsed_time(b) for a, b in lone) / len(lone)
    else:
        mine = [events[k] for k in range(args.steps) if plan(k, args.steps)[0] != "own" or plan(k, args.steps)[1] == rank]
        kernel_ms = sum(a.elapsed_time(b) for a, b in mine) / max(len(mine), 1)
    st = ctxs[lone_step % n_ctx].stats()
    tot = torch.tensor([dt, float(my_rays), kernel_ms], dtype=torch.float64, device=comm_dev)
    if dist_on:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, kernel_ms, rays_total = float(mx[0]), float(mx[2]), float(sm[1])
    else:
        rays_total = float(my_rays)

    # executed algorithmic flops of this rank's launch of step `lone_step`, from the counting build
    # of the same kernel on the same seed
    mode0, _ = plan(lone_step, args.steps)
    cam0, seed0 = wl.camera_of_step(lone_step)
    if wl.path:
        ctx.set_camera(**cam0)
    if mode0 == "tiles":
        cparams = P(W, H, SPP, DEPTH, seed0, tiles=tiles, flags=api.RT_FLAG_COUNTERS)
    elif mode0 == "samples":
        cparams = P(W, H, max(my_count, 1), DEPTH, seed0, sample_begin=my_begin, flags=api.RT_FLAG_COUNTERS)
    else:
        cparams = P(W, H, SPP, DEPTH, seed0, flags=api.RT_FLAG_COUNTERS)
    torch.cuda.synchronize()
    ctx.render(cparams, fbufs[0].data_ptr())
    cst = ctx.stats()
    flops = executed_flops(cst)
    brute = cst["rays"] * wl.bruteforce_flops_per_ray()
    per_ray = {k: cst[k] / max(cst["rays"], 1) for k in ("box_tests", "sphere_tests", "sphere_hits", "rect_tests", "wide_tests")}
    # traversal lane utilisation of the counting launch: lane steps / (64 x wave steps)
    trav_util = cst["phase_clocks"][0] / (64.0 * cst["phase_clocks"][1]) if cst["phase_clocks"][1] else None

    # the reference's own closest-hit algorithm (linear scan over all spheres,
    # sceneobject_list.h:26-39) on the same frame: its flop count per ray is fixed, which makes it
    # the like-for-like VALU roofline figure next to the BVH kernel's
    list_mode = None
    if world == 1 and args.accel == "bvh" and args.config == "c2":
        lctx = api.Context(device)
        lctx.set_stream(torch.cuda.current_stream().cuda_stream)
        lctx.upload_scene(wl.flat, accel=api.RT_ACCEL_LIST)
        lctx.set_camera(**wl.camera.desc())
        best = None
        for _ in range(3):
            lctx.render(P(W, H, SPP, DEPTH, wl.seed), fbufs[0].data_ptr())
            lst = lctx.stats()
            best = lst if best is None or lst["kernel_ms"] < best["kernel_ms"] else best
        lflops = best["rays"] * wl.bruteforce_flops_per_ray()
        list_mode = {"kernel_ms": best["kernel_ms"], "mrays_s": best["rays"] / best["kernel_ms"] / 1e3,
                     "achieved": lflops / (best["kernel_ms"] * 1e-3) / 1e12,
                     "frac": lflops / (best["kernel_ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                     "flops_per_ray": wl.bruteforce_flops_per_ray()}
        lctx.close()

    frame_ok = None
    if args.check:
        # each rank checks the frames it ended up holding against its own whole-frame render of the
        # same step; together the ranks must hold every step exactly once
        ok = True
        whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        out8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        for k, got in kept.items():
            cam, seed = wl.camera_of_step(k)
            ctx.set_camera(**cam)
            ctx.render(P(W, H, SPP, DEPTH, seed), whole.data_ptr())
            if wl.path:
                ctx.quantize(whole.data_ptr(), W * H, SPP, out8.data_ptr())
            ctx.stats()
            want = out8 if wl.path else whole
            ok = ok and bool(torch.equal(want.cpu(), got.cpu()))
        held = torch.tensor([len(kept), int(ok)], dtype=torch.int64, device=comm_dev)
        if dist_on:
            allh = [torch.zeros_like(held) for _ in range(world)]
            dist.all_gather(allh, held)
        else:
            allh = [held]
        frame_ok = all(int(h[1]) == 1 for h in allh) and sum(int(h[0]) for h in allh) == args.steps

    # the other configurations under the same run's clock (after the timed region, one launch at a
    # time): three frames each of c3 and c4 and one of the camera path, at their own sizes and
    # sample counts
    other = None
    if world == 1 and not dist_on and args.config == "c2" and args.accel == "bvh" and not args.no_other_configs and not args.spp:
        for c in ctxs:
            c.close()
        del fbufs
        torch.cuda.empty_cache()
        sp = torch.cuda.current_stream().cuda_stream
        other = {"c1": lone_frames("c1", device, sp, 3), "c3": lone_frames("c3", device, sp, 3), "c4": lone_frames("c4", device, sp, 3), "c5": lone_frames("c5", device, sp, 1),
                 "reconstruction": reconstruction_stage(device, sp)}

    if rank == 0:
        if not dist_on:
            how = "single GPU"
        elif partition == "tiles":
            how = f"row-tiles x{world}, RCCL gather to rank " + ("k % N (step k)" if args.gather_root == "rotate" and world > 1 else "0")
        elif partition == "samples":
            how = (f"sample-split x{world} as integer sums, RCCL int64 reduce to rank " +
                   ("k % N (step k)" if args.gather_root == "rotate" and world > 1 else "0") + " + resolve")
        else:
            how = (f"frames dealt in rounds of {world} ({full_rounds} of {args.steps} steps), the other "
                   f"{args.steps - full_rounds} sample-split x{world} (integer sums, int64 reduce)")
        out = {
            "metric": "Mrays/sec (whole node) + ms/frame 1080p@100spp",
            "value": rays_total / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.label, "config": args.config, "width": W, "height": H, "spp": SPP,
                       "accel": args.accel, "tile_rows": TILE_ROWS if (dist_on and partition == "tiles") else None,
                       "partition": how,
                       "rays_per_frame": rays_total / args.steps, "lanes_per_pixel": st["lanes_per_pixel"],
                       "bvh_depth": st["bvh_depth"], "bvh_nodes": st["bvh_nodes"], "bvh_coarse_nodes": st["bvh_coarse_nodes"], "bvh_build_ms": build_ms,
                       "frames_in_flight": args.frames_in_flight if args.steps > 1 else 1},
            "roofline": {"bound": "valu", "achieved": flops / (kernel_ms * 1e-3) / 1e12,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                         "traffic": measured_traffic(args.config) if world == 1 and args.accel == "bvh" else None,
                         "kernel": "render_kernel", "kernel_ms": kernel_ms, "frame_latency_ms": kernel_ms,
                         "one_launch_at_a_time_mrays_s": cst["rays"] / kernel_ms / 1e3,
                         "flops_per_launch": flops, "flops_per_ray": flops / max(cst["rays"], 1),
                         "tests_per_ray": per_ray, "traversal_lane_utilisation": trav_util,
                         "bruteforce_equivalent_tflops": brute / (kernel_ms * 1e-3) / 1e12,
                         "bruteforce_flops_per_ray": wl.bruteforce_flops_per_ray(),
                         "linear_scan_kernel": list_mode,
                         "note": "FP32 VALU-bound path (no dense contraction; HBM traffic is the framebuffer and, for c4, "
                                 "BVH nodes through L2); achieved = executed algorithmic flops of rank 0's launch "
                                 "(counting build of the same kernel, same seed) / kernel_ms, the HIP-event time of ONE "
                                 "launch running alone on the stream it is launched on.  value and ms_per_step are the "
                                 "throughput of the frame sequence: with two frames in flight a launch starts in the tail "
                                 "of the one before it, so ms_per_step is shorter than kernel_ms; the conservative "
                                 "one-launch-at-a-time rate is one_launch_at_a_time_mrays_s"},
        }
        out["roofline"]["memory_side"] = memory_side(args.config, cst, kernel_ms, W, H) if world == 1 else None
        if other is not None:
            out["other_configs"] = other
        if frame_ok is not None:
            out["frame_matches_single_gpu"] = frame_ok
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if args.config == "c2":
                out["speedup_vs_cpu_1thread"] = out["speedup_vs_cpu"]
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(wl)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
