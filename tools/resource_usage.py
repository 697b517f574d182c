#!/usr/bin/env python3
"""Register use of every render kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/resource_usage.py [extra hipcc flags]      (compiles rt_kernels.hip for gfx950; no GPU needed)

Columns are render_kernel's template arguments: COUNT LDS PROF MODE FEAT QUANT."""
import re
import subprocess
import sys

SRC = "raytrace_amd/csrc/rt_kernels.hip"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=on", "-fno-slp-vectorize"]


def main():
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", SRC, "-c", "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        m = re.search(r"render_kernelILb(\d)ELb(\d)ELb(\d)ELi(\d)ELi(\d+)ELi(\d)E", name)
        if not m:
            continue
        rows.append((tuple(int(x) for x in m.groups()), g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    print("COUNT LDS PROF MODE FEAT QUANT | VGPRs spilled sgpr-spilled scratch-bytes waves/SIMD")
    for key, v, sp, ss, sc, occ in sorted(rows):
        print("  %d    %d    %d    %d   %2d    %d   | %4d %5d %8d %10d %8d" % (*key, v, sp, ss, sc, occ))
    print(len(rows), "render kernels")


if __name__ == "__main__":
    main()
