"""Cuts one render kernel out of the device assembly and summarises it: instruction mix, scratch (spill)
traffic, and where in the kernel the scratch instructions sit (by loop nesting guessed from branch targets).
usage: hipcc ... --cuda-device-only -S -x hip rt_kernels.hip -o k.s ; python tools/isa_kernel.py k.s COUNT LDS PROF MODE FEAT QUANT [out.s]"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2:8]
tag = "render_kernelILb%sELb%sELb%sELi%sELi%sELi%sE" % tuple(key)
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and tag in l and ": " in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
if len(sys.argv) > 8:
    open(sys.argv[8], "w").write("\n".join(body))
ops = Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m and not l.strip().startswith((".", ";")):
        ops[m.group(1)] += 1
total = sum(ops.values())
print(tag, "instructions:", total)
for pre in ("v_", "s_", "ds_", "global_", "scratch_", "buffer_", "flat_"):
    print(f"  {pre:9s} {sum(v for k, v in ops.items() if k.startswith(pre))}")
print("  scratch:", {k: v for k, v in ops.items() if k.startswith("scratch_")})
print("  v_readlane/v_writelane:", ops["v_readlane_b32"], ops["v_writelane_b32"])
print("  top:", ops.most_common(14))
# labels and backward branches = loops
label_at = {}
for i, l in enumerate(body):
    m = re.match(r"(\.LBB[0-9_]+):", l)
    if m:
        label_at[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in label_at and label_at[t] < i:
            loops.append((label_at[t], i))
def depth(i):
    return sum(1 for a, b in loops if a <= i <= b)
sc = Counter()
for i, l in enumerate(body):
    if re.match(r"\s+scratch_(load|store)", l):
        sc[depth(i)] += 1
print("  scratch instructions by loop depth:", dict(sorted(sc.items())))
inner = sorted(loops, key=lambda ab: ab[1] - ab[0])
print("  loops (start line, length, scratch inside):", [(a, b - a, sum(1 for i in range(a, b + 1) if re.match(r"\s+scratch_", body[i]))) for a, b in inner[:14]])
