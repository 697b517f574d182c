"""Host time of rt_write_ppm for a 1920x1080 frame: the reference's P3 text (cuda/main.cu:358-383) and binary P6.
usage: python tools/ppm_probe.py"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                # noqa: E402
from raytrace_amd import api                      # noqa: E402

img = np.random.default_rng(1).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
with tempfile.TemporaryDirectory() as d:
    for binary in (False, True):
        ts = []
        for k in range(7):
            t0 = time.perf_counter()
            api.write_ppm(os.path.join(d, f"f{k}.ppm"), img, binary=binary)
            ts.append((time.perf_counter() - t0) * 1e3)
        size = os.path.getsize(os.path.join(d, "f0.ppm"))
        print(f"{'P6' if binary else 'P3'}: {min(ts):.1f} ms best, {sorted(ts)[len(ts) // 2]:.1f} ms median, {size / 1e6:.1f} MB")
