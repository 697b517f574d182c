"""Time and bandwidth of the reconstruction stage at 1920x1080 on the default world: guides, then the
a-trous passes on a 16-sample frame, against the 256-sample frame of the same camera (PSNR before and after).
usage: python tools/recon_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402


def psnr(img, ref, spp_img, spp_ref):
    """PSNR of the displayed frames (mean radiance through gamma 2 and the clamp of color.h:17-23, peak 1)."""
    shown = lambda x, n: torch.sqrt(torch.clamp(x[..., :3] / n, 0.0, 1.0))
    return float(-10.0 * torch.log10(torch.mean((shown(img, spp_img) - shown(ref, spp_ref)) ** 2)))


W, H = 1920, 1080
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()))
ctx.set_camera(**S.rtiow_camera(float(np.float32(W / H))).desc())
low = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
high = torch.zeros_like(low)
out = torch.zeros_like(low)
g = torch.zeros((2, H, W, 4), dtype=torch.float32, device="cuda")
ctx.render(api.Context.params(W, H, 256, 50, 1), high.data_ptr()); ctx.stats()
for spp in (4, 16):
    ctx.render(api.Context.params(W, H, spp, 50, 2), low.data_ptr())
    t_low = ctx.stats()["kernel_ms"]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for rep in range(3):
        torch.cuda.synchronize()
        ev[0].record(); ctx.render_guides(W, H, g.data_ptr()); ev[1].record()
        ctx.reconstruct(low.data_ptr(), g.data_ptr(), W, H, out.data_ptr()); ev[2].record()
        torch.cuda.synchronize()
    t_g, t_r = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    for rep in range(3):
        torch.cuda.synchronize()
        ev[0].record(); ctx.reconstruct(low.data_ptr(), g.data_ptr(), W, H, out.data_ptr(), flags=api.RT_RECON_NO_LDS); ev[1].record()
        torch.cuda.synchronize()
    t_direct = ev[0].elapsed_time(ev[1])
    passes = 4
    algo = W * H * (48 + 16 + (passes - 1) * (36 + 16))          # first pass 48 B in + 16 out (demodulation folded in), then 36 in + 16 out per pass
    print(f"{spp:3d} spp frame {t_low:7.3f} ms | guides {t_g:6.3f} ms | reconstruct ({passes} passes) {t_r:6.3f} ms = "
          f"{algo / t_r / 1e6:7.1f} GB/s of algorithmic bytes (every pass through the caches: {t_direct:6.3f} ms) | PSNR against 256 spp: {psnr(low, high, spp, 256):.2f} -> "
          f"{psnr(out, high, spp, 256):.2f} dB", flush=True)
ctx.close()
