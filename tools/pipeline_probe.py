"""Frames per second of one rank's shard when consecutive frames are launched on two streams
(two contexts, two buffers), so that a frame starts in the tail of the one before it, against
one stream."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytrace_amd import api, scene as S
aspect = float(np.float32(16 / 9))
flat = S.flatten(S.rtiow_scene())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs = []
for s in streams:
    c = api.Context(0); c.set_stream(s.cuda_stream); c.upload_scene(flat); c.set_camera(**S.rtiow_camera(aspect).desc()); ctxs.append(c)
bufs = [torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
N = 40
KS = [int(k) for k in os.environ.get("RT_PROBE_K", "0").split(",")]
for world, K in [(w, k) for w in (1, 2, 4, 8) for k in KS]:
    p = api.Context.params(1920, 1080, 100, 50, 1, tiles=(4, 0, world) if world > 1 else (0, 0, 0), lanes_per_pixel=K)
    for mode in ("one stream", "two streams"):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(N):
                i = (k & 1) if mode == "two streams" else 0
                ctxs[i].render(p, bufs[i].data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N * 1e3
        print(f"world {world} K={K} {mode}: {dt:.3f} ms/frame", flush=True)
ref = bufs[0].clone()
print("same bits on both streams:", bool(torch.equal(bufs[0], bufs[1])))
