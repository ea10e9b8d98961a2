#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a short markdown + JSON summary: per-launch numbers of ONE kernel
(name fragment = third argument or $KERN, default "lnprob_kernel"; e.g. stretch_kernel, lnprob_pc_kernel)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
KERN = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("KERN", "lnprob_kernel")


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


summary = {"tag": tag}
f = find("trace", "*kernel_stats.csv")
lines = ["# rocprofv3 summary `%s`" % tag, ""]
if f:
    lines += ["## kernel-trace --stats (top kernels)", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for row in list(csv.DictReader(open(f)))[:5]:
        name = row["Name"]
        short = name.split("(")[0][-60:] if len(name) > 70 else name
        lines.append(f"| `{short}` | {row['Calls']} | {float(row['AverageNs'])/1e3:.1f} | {float(row['MinNs'])/1e3:.1f} | "
                     f"{float(row['MaxNs'])/1e3:.1f} | {float(row['Percentage']):.2f} |")
        if KERN in name and "kernel_avg_us" not in summary:
            summary.update(kernel=name.split("(")[0], calls=int(row["Calls"]), kernel_avg_us=float(row["AverageNs"]) / 1e3,
                           kernel_min_us=float(row["MinNs"]) / 1e3)
    lines.append("")
counters = defaultdict(list)
meta = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mix", "pmc_misc"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    per_dispatch = defaultdict(dict)
    for row in csv.DictReader(open(f)):
        if KERN not in row.get("Kernel_Name", ""):
            continue
        per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
        meta = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Grid_Size",
                                        "Workgroup_Size", "Scratch_Size") if k in row}
    for d in per_dispatch.values():
        for k, v in d.items():
            counters[k].append(v)
avg = {k: sum(v) / len(v) for k, v in counters.items()}
summary["counters_per_launch"] = avg
summary["dispatch"] = meta
if avg:
    lines += ["## PMC counters, average per launch of `mp::%s`" % KERN, "", "| counter | value |", "|---|---|"]
    for k in sorted(avg):
        lines.append(f"| {k} | {avg[k]:.6g} |")
    lines.append("")
    d = {}
    if "SQ_WAVES" in avg and avg["SQ_WAVES"]:
        w = avg["SQ_WAVES"]
        d["valu_insts_per_wave"] = avg.get("SQ_INSTS_VALU", 0) / w
        d["salu_insts_per_wave"] = avg.get("SQ_INSTS_SALU", 0) / w
        if avg.get("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md cycle constants)
            d["wave_cycles_per_wave"] = 4.0 * avg["SQ_WAVE_CYCLES"] / w
            d["cycles_per_valu_inst"] = 4.0 * avg["SQ_WAVE_CYCLES"] / max(avg.get("SQ_INSTS_VALU", 1), 1)
            d["valu_active_frac_of_wave_cycles"] = avg.get("SQ_ACTIVE_INST_VALU", 0) / avg["SQ_WAVE_CYCLES"]
            d["wait_inst_any_frac"] = avg.get("SQ_WAIT_INST_ANY", 0) / avg["SQ_WAVE_CYCLES"]
            d["wait_any_frac"] = avg.get("SQ_WAIT_ANY", 0) / avg["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU_FMA_F64" in avg and "SQ_WAVES" in avg:
        w = avg["SQ_WAVES"]
        f64 = avg["SQ_INSTS_VALU_FMA_F64"] + avg.get("SQ_INSTS_VALU_MUL_F64", 0) + avg.get("SQ_INSTS_VALU_ADD_F64", 0)
        d["f64_arith_insts_per_wave"] = f64 / w
        d["f64_trans_insts_per_wave"] = avg.get("SQ_INSTS_VALU_TRANS_F64", 0) / w
        # flops actually executed: FMA = 2, MUL/ADD = 1 per lane, 64 lanes
        flops = 64.0 * (2.0 * avg["SQ_INSTS_VALU_FMA_F64"] + avg.get("SQ_INSTS_VALU_MUL_F64", 0) + avg.get("SQ_INSTS_VALU_ADD_F64", 0))
        d["fp64_flops_per_launch"] = flops
        if "kernel_avg_us" in summary:
            d["fp64_tflops_achieved"] = flops / (summary["kernel_avg_us"] * 1e-6) / 1e12
            d["fp64_valu_frac_of_78.6TF"] = d["fp64_tflops_achieved"] / 78.6
    if "FETCH_SIZE" in avg:
        d["hbm_read_bytes_raw"] = avg["FETCH_SIZE"] * 1024.0          # FETCH_SIZE is in KiB
        d["hbm_read_bytes_x2_gfx950"] = 2.0 * avg["FETCH_SIZE"] * 1024.0  # guide: gfx950 tallies 128-B requests at 64 B
    if "WRITE_SIZE" in avg:
        d["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes_raw" in d and "hbm_write_bytes" in d:
        d["traffic_bytes_per_launch"] = d["hbm_read_bytes_x2_gfx950"] + d["hbm_write_bytes"]
        if "kernel_avg_us" in summary:
            d["hbm_write_GBps"] = d["hbm_write_bytes"] / (summary["kernel_avg_us"] * 1e-6) / 1e9
            d["hbm_traffic_GBps"] = d["traffic_bytes_per_launch"] / (summary["kernel_avg_us"] * 1e-6) / 1e9
    summary["derived"] = d
    lines += ["## derived", "", "| quantity | value |", "|---|---|"]
    for k, v in d.items():
        lines.append(f"| {k} | {v:.6g} |")
    if meta:
        lines += ["", "dispatch: " + ", ".join(f"{k}={v}" for k, v in meta.items())]
        # LDS occupancy (BASELINE.json configs[2] asks for it): resident one-wave workgroups per CU x LDS bytes per workgroup
        # against the 160 KB of a CU.  Waves per SIMD are fixed by the kernels' amdgpu_waves_per_eu attribute (mp_kernels.hip:
        # one for the 4-steps-per-lane builds, which take a SIMD's whole register file, two for the 2-steps-per-lane ones);
        # the register fields of the dispatch record are not used for this.
        try:
            import re
            lds = int(meta.get("LDS_Block_Size") or 0)
            wg = int(meta.get("Workgroup_Size") or 64)
            m = re.search(r"(?:lnprob_kernel<(?:true|false), (\d)|stretch(?:_step)?_kernel<(\d))", str(summary.get("kernel", "")))
            spl = int(next(g for g in m.groups() if g)) if m else 0
            if spl and lds > 0 and wg == 64:
                waves_simd = 1 if spl >= 4 else 2
                by_regs = 4 * waves_simd
                by_lds = (160 * 1024) // lds
                resident = min(by_regs, by_lds)
                occ = {"waves_per_simd": waves_simd, "workgroups_per_cu_by_registers": by_regs, "workgroups_per_cu_by_lds": by_lds,
                       "resident_workgroups_per_cu": resident, "lds_bytes_in_use_per_cu": resident * lds,
                       "lds_occupancy_frac_of_160KB": resident * lds / (160.0 * 1024.0),
                       "limited_by": "registers" if by_regs <= by_lds else "LDS"}
                summary["lds_occupancy"] = occ
                lines += ["", f"LDS occupancy: {lds} B per one-wave workgroup x {resident} resident workgroups per CU ({waves_simd} "
                              f"per SIMD, limited by {occ['limited_by']}; LDS alone would allow {by_lds}) = {resident * lds / 1024.0:.1f} KB "
                              f"of the CU's 160 KB ({100.0 * occ['lds_occupancy_frac_of_160KB']:.0f} %)"]
        except (TypeError, ValueError, StopIteration):
            pass
for b in ("bench_trace.json",):
    p = os.path.join(out, b)
    if os.path.exists(p):
        try:
            j = json.loads(open(p).read().strip().splitlines()[-1])
            summary["bench_under_profiler"] = {"value": j["value"], "kernel_ms_avg": j["roofline"]["kernel_ms_avg"],
                                               "workload": j["config"]["workload"],
                                               "evals_per_launch": j["roofline"]["evals_per_launch"],
                                               "ensemble_sampler": j.get("ensemble_sampler")}
            lines += ["", f"bench.py under the profiler: {j['value']:.0f} evals/s, HIP-event kernel time "
                          f"{j['roofline']['kernel_ms_avg']*1e3:.1f} us ({j['config']['workload']})"]
        except Exception as e:  # noqa: BLE001
            lines += ["", f"(bench json unreadable: {e})"]
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("\n".join(lines))
