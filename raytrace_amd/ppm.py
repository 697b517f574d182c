"""Frame output in the reference's formats: plain-text P3 PPM, rows top to bottom, one pixel per
line (main.cc:109-126 with color.h:23-25; cuda/main.cu:358-383), frame names as
cuda/main.cu:569."""
from __future__ import annotations

import numpy as np


def p3_bytes(rgb8_bottom_up: np.ndarray) -> bytes:
    """rgb8_bottom_up: [H, W, 3] uint8 with row 0 = bottom row (the framebuffer orientation).
    The text is laid out with array operations (a 1080p frame is six million numbers: a Python
    loop over them takes longer than a hundred frames take to render)."""
    h, w, _ = rgb8_bottom_up.shape
    v = np.ascontiguousarray(rgb8_bottom_up[::-1]).reshape(-1).astype(np.int64)
    n_digits = 1 + (v >= 10) + (v >= 100)
    end = np.cumsum(n_digits + 1)                           # one past each number's separator
    out = np.empty(int(end[-1]) if v.size else 0, np.uint8)
    sep = np.full(v.size, ord(" "), np.uint8)
    sep[2::3] = ord("\n")                                   # one pixel per line
    out[end - 1] = sep
    out[end - 2] = ord("0") + v % 10
    tens = v >= 10
    out[end[tens] - 3] = ord("0") + (v[tens] // 10) % 10
    hundreds = v >= 100
    out[end[hundreds] - 4] = ord("0") + v[hundreds] // 100
    return f"P3\n{w} {h}\n255\n".encode("ascii") + out.tobytes()


def write_p3(path: str, rgb8_bottom_up: np.ndarray) -> None:
    """The file p3_bytes describes, written by the library (rt_write_ppm: 5 ms of host time for a 1080p frame)."""
    from . import api
    api.write_ppm(path, rgb8_bottom_up, binary=False)


def write_p6(path: str, rgb8_bottom_up: np.ndarray) -> None:
    from . import api
    api.write_ppm(path, rgb8_bottom_up, binary=True)


def frame_name(frame: int, quality: str = "low") -> str:
    return f"image_{frame}_{quality}.ppm"          # cuda/main.cu:569,575
