"""Turns the output of tools/profile_round.sh into the small files kept under profiles/.

usage: python tools/profile_summarize.py gpurun_out/<dir> profiles/<round>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "c2_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_traced.json")) if l.startswith("{")]
if line:
    with open(os.path.join(dst, "c2_bench.json"), "w") as f:
        f.write(line[-1])

per_kernel = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "render_kernel" not in name:
            continue
        # one row per dispatch, counter and counter dimension: add up a dispatch's rows
        per_kernel[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {}
for name, ctrs in per_kernel.items():
    out[name] = {k: sum(v.values()) / len(v) for k, v in sorted(ctrs.items())}
    out[name]["launches_per_pass"] = max(len(v) for v in ctrs.values())
with open(os.path.join(dst, "c2_pmc_per_launch.json"), "w") as f:
    json.dump(out, f, indent=1)

main = [n for n in out if "render_kernel<false, true, false, 1, false>" in n]      # plain, LDS-resident BVH, no rects
if main:
    c = out[main[0]]
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    traffic = {
        "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline ; "
                   "same with --pmc WRITE_SIZE (separate passes)",
        "kernel": main[0],
        "FETCH_SIZE_KB_per_launch": fetch_kb,
        "WRITE_SIZE_KB_per_launch": write_kb,
        "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); immaterial next to the writes",
        "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
        "algorithmic_bytes_per_launch": 1920 * 1080 * 16,
        "note": "framebuffer float4 stores in 64-B row segments of the 4x2 wave tile; the scene (30 KB) is "
                "read once per workgroup into LDS",
    }
    with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))
print(json.dumps(out, indent=1))
