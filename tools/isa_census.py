#!/usr/bin/env python3
"""Static instruction census per kernel of mp_kernels.hip (hipcc -S): totals by class, barriers, spill moves.

    python tools/isa_census.py [name filter] [extra hipcc flags ...]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
out = "/tmp/mp_kernels_census.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
                os.path.join(ROOT, "magprop_amd", "csrc", "mp_kernels.hip"), "-o", out, *sys.argv[2:]], check=True,
               stderr=subprocess.DEVNULL)
s = open(out).read()
for f in re.split(r"\n(?=_Z\w+:)", s):
    name = f.split(":")[0]
    if not name.startswith("_Z") or flt not in name or "kernel" not in name:
        continue
    body = f.split(".Lfunc_end")[0]
    c = collections.Counter(re.findall(r"^\s+([a-z][a-z_0-9]+)", body, re.M))
    def tot(pred):
        return sum(v for k, v in c.items() if pred(k))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("mp::", "").replace("void ", "")
    print(f"{dem:48s} valu {tot(lambda k: k.startswith('v_')):6d} (f64 {tot(lambda k: k.endswith('_f64')):6d}) salu {tot(lambda k: k.startswith('s_')):6d} "
          f"ds {tot(lambda k: k.startswith('ds_')):5d} flat {tot(lambda k: k.startswith('flat_')):3d} global {tot(lambda k: k.startswith('global_')):4d} "
          f"scratch {tot(lambda k: k.startswith('scratch_')):3d} barrier {c.get('s_barrier', 0):3d} accvgpr {tot(lambda k: 'accvgpr' in k):5d} "
          f"rd/wrlane {c.get('v_readlane_b32', 0) + c.get('v_writelane_b32', 0):5d} dpp {len(re.findall(r'_dpp', body)):4d} waitcnt {c.get('s_waitcnt', 0):5d}")
