#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/ and rebuild profiles/pmc_figures.json,
the per-launch PMC figures bench.py reads (keyed by mode and walkers per launch).
    python tools/collect_profiles.py r02_n1024:lnprob r02_n4096:lnprob r02_curve1024:curve r02_stretch512:stretch ..."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fig_path = os.path.join(ROOT, "profiles", "pmc_figures.json")
fig = json.load(open(fig_path)) if os.path.exists(fig_path) else {}
fig["_comment"] = ("Per-launch figures of the hot-path kernels from rocprofv3 PMC passes (tools/profile.sh; summaries in "
                   "profiles/<tag>_rocprof_summary.{md,json}), keyed by mode and walkers (or proposals) per launch. "
                   "traffic = FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, bytes; fp64_flops = 64 "
                   "lanes x (2 FMA + MUL + ADD) instructions.")
for spec in sys.argv[1:]:
    tag, mode = spec.split(":")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    s = json.load(open(os.path.join(src, "summary.json")))
    for ext in ("md", "json"):
        shutil.copy(os.path.join(src, "summary." + ext), os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.{ext}"))
    d = s.get("derived", {})
    n = str(int(s["counters_per_launch"].get("SQ_WAVES", 0)))
    if mode not in ("stretch",) and "bench_under_profiler" in s:
        n = str(s["bench_under_profiler"]["evals_per_launch"])
    fig.setdefault(mode, {})[n] = {
        "traffic_bytes": d.get("traffic_bytes_per_launch"), "fp64_flops": d.get("fp64_flops_per_launch"),
        "kernel": s.get("kernel"), "kernel_avg_us": s.get("kernel_avg_us"), "vgprs": s.get("dispatch", {}).get("VGPR_Count"),
        "lds_bytes": s.get("dispatch", {}).get("LDS_Block_Size"), "scratch": s.get("dispatch", {}).get("Scratch_Size"),
        "valu_insts_per_wave": d.get("valu_insts_per_wave"), "salu_insts_per_wave": d.get("salu_insts_per_wave"),
        "f64_arith_insts_per_wave": d.get("f64_arith_insts_per_wave"), "wave_cycles_per_wave": d.get("wave_cycles_per_wave"),
        "fp64_tflops": d.get("fp64_tflops_achieved"), "hbm_write_GBps": d.get("hbm_write_GBps"), "source": f"profiles/{tag}_rocprof_summary.json"}
    print(tag, mode, n, fig[mode][n])
sys.path.insert(0, ROOT)
import bench  # noqa: E402
fig["build"] = bench.csrc_hash()      # the sources these counters belong to (bench.py reports `stale` on a mismatch)
json.dump(fig, open(fig_path, "w"), indent=1)
