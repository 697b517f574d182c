"""Multi-frame camera-path driver: the frame loop of the reference's CUDA host program
(cuda/main.cu:557-591) on the GPU path, plus the post-processing its helper scripts shell out for.

    for frame: update_scene (cuda/main.cu:220-279) -> render -> write_color -> image_<frame>_low.ppm

`render_sequence` writes P3 text like the reference (cuda/main.cu:358-383, with color.h's clamp),
binary P6, or PNG directly, which replaces cuda/output/convert.py (ImageMagick per file).
`render_pairs` writes the low / high sample-count pair of every frame that loop is laid out for.
`encode_video` is cuda/output/generate_video.py: the same ffmpeg command line, run only when
ffmpeg exists.  With several ranks (one process per GPU) frames are dealt round-robin: frames are
independent, so there is nothing to exchange.
"""
from __future__ import annotations

import os
import shutil
import struct
import subprocess
import zlib

import numpy as np

from . import ppm
from . import scene as S


def png_bytes(rgb8_bottom_up: np.ndarray) -> bytes:
    """Minimal PNG encoder (8-bit RGB, no filtering): rows top to bottom."""
    h, w, _ = rgb8_bottom_up.shape
    rows = np.ascontiguousarray(rgb8_bottom_up[::-1]).reshape(h, w * 3)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def frame_path(out_dir: str, frame: int, fmt: str, quality: str = "low") -> str:
    if fmt == "png":
        # cuda/output/convert.py:17-30 -> png_images/<quality>/image_<n>.png
        return os.path.join(out_dir, "png_images", quality, f"image_{frame}.png")
    return os.path.join(out_dir, "ppm_images", ppm.frame_name(frame, quality))   # cuda/main.cu:569


def render_sequence(ctx, cam: S.camera, frames, width: int, height: int, spp: int, out_dir: str | None,
                    max_depth: int = 25, seed: int = 1, fmt: str = "p3", quality: str = "low",
                    rank: int = 0, world: int = 1, first_frame: int = 0, overlap_ctx=None):
    """Renders frames first_frame .. first_frame+frames-1 of the camera path that starts at `cam`.
    Every rank advances the camera through all frames (cheap, host) and renders those with
    frame % world == rank.  Returns [(frame, rays, kernel_ms)] of the frames this rank rendered.

    overlap_ctx: a second context holding the same scene.  Frames then alternate between the two
    contexts and a frame is launched before the one before it is read back and written, so the
    GPU renders while the host formats the file (P3 text takes longer to write than a frame takes
    to render); with the contexts on two streams the frame also starts in the tail of the one
    before it.  The files are the same either way."""
    import torch

    from . import api

    if fmt not in ("p3", "p6", "png"):
        raise ValueError(f"unknown format {fmt!r}")
    ctxs = [ctx] if overlap_ctx is None else [ctx, overlap_ctx]
    fbs = [torch.zeros((height, width, 4), dtype=torch.float32, device=f"cuda:{c.device}") for c in ctxs]
    rgbs = [torch.zeros((height, width, 3), dtype=torch.uint8, device=f"cuda:{c.device}") for c in ctxs]
    torch.cuda.synchronize()                                    # the buffers are zeroed on torch's stream
    done = []

    def finish(frame, slot):
        st = ctxs[slot].stats()                                 # waits for that context's stream
        img = rgbs[slot].cpu().numpy()
        if out_dir is not None:
            path = frame_path(out_dir, frame, fmt, quality)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            if fmt == "p3":
                ppm.write_p3(path, img)
            elif fmt == "p6":
                ppm.write_p6(path, img)
            else:
                with open(path, "wb") as f:
                    f.write(png_bytes(img))
        done.append((frame, st["rays"], st["kernel_ms"]))

    pending, launched = None, 0
    for frame in range(first_frame + frames):
        S.update_scene(cam, frame)                              # cuda/main.cu:564
        if frame < first_frame or frame % world != rank:
            continue
        slot = launched % len(ctxs)
        launched += 1
        c = ctxs[slot]
        c.set_camera(**cam.desc())
        c.render(api.Context.params(width, height, spp, max_depth, seed=seed + frame), fbs[slot].data_ptr())
        c.quantize(fbs[slot].data_ptr(), width * height, spp, rgbs[slot].data_ptr())
        if pending is not None:
            finish(*pending)
        pending = (frame, slot)
        if len(ctxs) == 1:
            finish(*pending)
            pending = None
    if pending is not None:
        finish(*pending)
    return done


def render_pairs(ctx, cam: S.camera, frames, width: int, height: int, low_spp: int, high_spp: int, out_dir: str | None,
                 max_depth: int = 25, seed: int = 1, fmt: str = "p3", rank: int = 0, world: int = 1, first_frame: int = 0,
                 reconstruct: bool = False):
    """The reference's "low-high samples of each image" loop (cuda/main.cu:557-583: image_<frame>_low at
    low_num_samples, image_<frame>_high at high_num_samples, main.cu:445-446, :569, :575) for the cost of
    the high frames alone.  Samples are keyed by their index, so the low frame is the first low_spp
    samples of the high one: each frame is rendered as two windows of samples into integer sums, the low
    file is written from the first window and the high file from the two windows added (integer sums
    add exactly: the high file is bit for bit the one a single render at high_spp writes).
    Returns [(frame, rays, kernel_ms)] with both windows counted.

    reconstruct: also turn every low frame into an estimate of its high frame (rt_render_guides +
    rt_reconstruct: the stage the reference's README.md:7 names as ongoing work and these pairs are laid
    out for), write it as image_<frame>_recon, and return [(frame, rays, kernel_ms, psnr_low, psnr_recon)]:
    the peak signal-to-noise ratio of the displayed low and reconstructed frames against the high one."""
    import torch

    from . import api

    if not 1 <= low_spp < high_spp:
        raise ValueError("need 1 <= low_spp < high_spp")
    if fmt not in ("p3", "p6", "png"):
        raise ValueError(f"unknown format {fmt!r}")
    dev = f"cuda:{ctx.device}"
    sums = [torch.zeros((height, width, 4), dtype=torch.int64, device=dev) for _ in range(2)]   # 32.32 sums as their bit patterns
    fb = torch.zeros((height, width, 4), dtype=torch.float32, device=dev)
    rgb = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev)
    guides = torch.zeros((2, height, width, 4), dtype=torch.float32, device=dev) if reconstruct else None
    shown = {}
    torch.cuda.synchronize()
    done = []

    def write(frame, quality, spp, sums_t):
        ctx.resolve_sums(sums_t.data_ptr(), width * height, fb.data_ptr())
        if quality == "recon":
            ctx.render_guides(width, height, guides.data_ptr())
            ctx.reconstruct(fb.data_ptr(), guides.data_ptr(), width, height, fb.data_ptr())
        if reconstruct:                                         # the displayed frame: color.h:17-23 without the bytes
            ctx.stats()
            shown[quality] = torch.sqrt(torch.clamp(fb[..., :3] / spp, 0.0, 1.0))
        ctx.quantize(fb.data_ptr(), width * height, spp, rgb.data_ptr())
        ctx.stats()                                             # waits for the context's stream
        img = rgb.cpu().numpy()
        if out_dir is not None:
            path = frame_path(out_dir, frame, fmt, quality)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            if fmt == "p3":
                ppm.write_p3(path, img)
            elif fmt == "p6":
                ppm.write_p6(path, img)
            else:
                with open(path, "wb") as f:
                    f.write(png_bytes(img))

    for frame in range(first_frame + frames):
        S.update_scene(cam, frame)                              # cuda/main.cu:564
        if frame < first_frame or frame % world != rank:
            continue
        ctx.set_camera(**cam.desc())
        rays, ms = 0, 0.0
        for k, (begin, n) in enumerate(((0, low_spp), (low_spp, high_spp - low_spp))):
            ctx.render(api.Context.params(width, height, n, max_depth, seed=seed + frame, sample_begin=begin,
                                          flags=api.RT_FLAG_FIXED_SUMS), sums[k].data_ptr())
            st = ctx.stats()
            rays, ms = rays + st["rays"], ms + st["kernel_ms"]
        write(frame, "low", low_spp, sums[0])
        if reconstruct:
            write(frame, "recon", low_spp, sums[0])
        sums[1] += sums[0]                                      # (on torch's stream; the context's is idle: write() waited)
        torch.cuda.synchronize()
        write(frame, "high", high_spp, sums[1])
        if reconstruct:
            psnr = {q: float(-10.0 * torch.log10(torch.mean((shown[q] - shown["high"]) ** 2))) for q in ("low", "recon")}
            done.append((frame, rays, ms, psnr["low"], psnr["recon"]))
        else:
            done.append((frame, rays, ms))
    return done


def encode_video(out_dir: str, fps: int = 60, quality: str = "low", name: str = "render_video.mp4"):
    """cuda/output/generate_video.py:2-12.  Returns the command; runs it only when ffmpeg exists."""
    cmd = ["ffmpeg", "-r", str(fps), "-i", os.path.join(out_dir, "png_images", quality, "image_%01d.png"),
           "-c:v", "libx264", "-crf", "20", "-pix_fmt", "yuv420p", os.path.join(out_dir, name)]
    if shutil.which("ffmpeg"):
        subprocess.run(cmd, check=True)
        return cmd, True
    return cmd, False
