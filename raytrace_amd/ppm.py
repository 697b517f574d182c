"""Frame output in the reference's formats: plain-text P3 PPM, rows top to bottom, one pixel per
line (main.cc:109-126 with color.h:23-25; cuda/main.cu:358-383), frame names as
cuda/main.cu:569."""
from __future__ import annotations

import numpy as np


def p3_bytes(rgb8_bottom_up: np.ndarray) -> bytes:
    """rgb8_bottom_up: [H, W, 3] uint8 with row 0 = bottom row (the framebuffer orientation)."""
    h, w, _ = rgb8_bottom_up.shape
    top_down = rgb8_bottom_up[::-1].reshape(-1, 3)
    body = "\n".join(f"{r} {g} {b}" for r, g, b in top_down.tolist())
    return f"P3\n{w} {h}\n255\n{body}\n".encode("ascii")


def write_p3(path: str, rgb8_bottom_up: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(p3_bytes(rgb8_bottom_up))


def write_p6(path: str, rgb8_bottom_up: np.ndarray) -> None:
    h, w, _ = rgb8_bottom_up.shape
    with open(path, "wb") as f:
        f.write(f"P6\n{w} {h}\n255\n".encode("ascii"))
        f.write(np.ascontiguousarray(rgb8_bottom_up[::-1]).tobytes())


def frame_name(frame: int, quality: str = "low") -> str:
    return f"image_{frame}_{quality}.ppm"          # cuda/main.cu:569,575
