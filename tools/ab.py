"""Same-box A/B of two builds of the library on the scenes of tools/regress.py: the cases alternate between the
two libraries (a process each per case would pay a start-up per case, so each library gets one process and the
two run one after the other; the second is not favoured: run it both ways round with --swap).
usage: python tools/ab.py build_variants/librt_A.so [raytrace_amd/librt_amd.so] [--quick]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
a = args[0]
b = args[1] if len(args) > 1 else os.path.join(ROOT, "raytrace_amd", "librt_amd.so")
quick = ["--quick"] if "--quick" in sys.argv else []
out = {}
for name, lib in (("A", a), ("B", b), ("A2", a), ("B2", b)):
    path = f"/tmp/ab_{name}.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "regress.py"), "--lib", lib] + quick + [path], check=True, stdout=subprocess.DEVNULL)
    out[name] = json.load(open(path))
print(f"A = {a}\nB = {b}\n{'case':58s} {'A ms':>9s} {'B ms':>9s}  B/A    same bits")
for ra, rb, ra2, rb2 in zip(out["A"], out["B"], out["A2"], out["B2"]):
    ma, mb = min(ra["kernel_ms"], ra2["kernel_ms"]), min(rb["kernel_ms"], rb2["kernel_ms"])
    print(f"{ra['case']:58s} {ma:9.3f} {mb:9.3f}  {mb / ma:5.3f}  {ra['digest'] == rb['digest'] and ra['rays'] == rb['rays']}")
