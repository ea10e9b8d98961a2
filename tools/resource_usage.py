#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in mp_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "magprop_amd", "csrc")


def report(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", "mp_kernels.hip",
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
        if m and "Name:" in line:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(r"\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


if __name__ == "__main__":
    rows = report(sys.argv[1:])
    print(f"{'kernel':58s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sspill':>7s} {'vspill':>7s} {'LDS':>7s}")
    for r in rows:
        name = re.sub(r"\(.*", "", r["name"]).replace("mp::", "").replace("void ", "")
        print(f"{name:58s} {r.get('TotalSGPRs', -1):5d} {r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('ScratchSize', -1):8d} "
              f"{r.get('Occupancy', -1):4d} {r.get('SGPRs Spill', -1):7d} {r.get('VGPRs Spill', -1):7d} {r.get('LDS Size', -1):7d}")
