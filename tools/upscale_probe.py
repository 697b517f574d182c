"""Low-resolution frames of the default world brought to 1920x1080 against the 1920x1080 @256 frame: plain
bilinear interpolation, rt_upscale, and rt_reconstruct followed by rt_upscale (PSNR of the displayed frames).
usage: python tools/upscale_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
import torch                                      # noqa: E402
from raytrace_amd import api, scene as S          # noqa: E402

W, H = 1920, 1080
ctx = api.Context(0)
ctx.upload_scene(S.flatten(S.rtiow_scene()))
ctx.set_camera(**S.rtiow_camera(float(np.float32(W / H))).desc())
Z = lambda *s: torch.zeros(s, dtype=torch.float32, device="cuda")
high, gh, out = Z(H, W, 4), Z(2, H, W, 4), Z(H, W, 4)
ctx.render(api.Context.params(W, H, 256, 50, 1), high.data_ptr())
ctx.render_guides(W, H, gh.data_ptr())
ctx.stats()
shown = lambda x, n: torch.sqrt(torch.clamp(x[..., :3] / n, 0.0, 1.0))
psnr = lambda a, n: float(-10.0 * torch.log10(torch.mean((shown(a, n) - shown(high, 256)) ** 2)))
for f in (2, 3):
    lw, lh = W // f, H // f
    low, den, gl = Z(lh, lw, 4), Z(lh, lw, 4), Z(2, lh, lw, 4)
    for spp in (4, 16, 64):
        ctx.render(api.Context.params(lw, lh, spp, 50, 2), low.data_ptr())
        t_low = ctx.stats()["kernel_ms"]
        ctx.render_guides(lw, lh, gl.data_ptr())
        ctx.reconstruct(low.data_ptr(), gl.data_ptr(), lw, lh, den.data_ptr())
        ctx.stats()
        res = {}
        for name, src in (("raw", low), ("denoised", den)):
            bil = torch.nn.functional.interpolate((src[..., :3] / spp).permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
            res["bilinear " + name] = psnr(bil * spp, spp)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            ctx.upscale(src.data_ptr(), gl.data_ptr(), lw, lh, gh.data_ptr(), W, H, out.data_ptr())
            ev[1].record()
            ctx.stats()
            torch.cuda.synchronize()
            res["rt_upscale " + name] = psnr(out, spp)
        print(f"1/{f} resolution ({lw}x{lh}) @{spp:<3d} {t_low:6.2f} ms | " + " | ".join(f"{k} {v:.2f} dB" for k, v in res.items()) + f" | upscale {ev[0].elapsed_time(ev[1]):.3f} ms", flush=True)
ctx.close()
