#!/usr/bin/env python3
"""Summarises a tools/prof_pmc.sh output directory: per kernel name, sums of every counter and the
kernel-trace durations.  usage: tools/pmc_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
short = lambda n: n.split("(")[0].replace("void ", "").replace("rt::", "")
dur = defaultdict(lambda: [0, 0])
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k][0] += 1
        dur[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("== kernel trace: name calls total_ms avg_us")
for k, (c, t) in sorted(dur.items(), key=lambda x: -x[1][1]):
    print(f"{k:40s} {c:6d} {t / 1e6:10.3f} {t / c / 1e3:10.1f}")
cnt = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
print("== PMC sums over all dispatches of each kernel")
for k in sorted(cnt, key=lambda k: -dur.get(k, [0, 0])[1]):
    print(k)
    for c, v in sorted(cnt[k].items()):
        print(f"    {c:28s} {v:.6g}")
# Derived figures.  SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md constants table); GRBM_GUI_ACTIVE
# is summed over the 8 XCDs; 1024 SIMDs.  VALU utilisation = SIMD-cycles spent issuing vector ALU work / SIMD-cycles the kernel ran.
# A wave's VALU-active interval can overlap another wave's on the same SIMD (multi-cycle f64 operations in the pipeline), so the
# first figure is VALU-active wave-cycles per SIMD-cycle and may exceed 1; it is the utilisation figure the kernels are steered by.
print("== derived per kernel: valu_active_per_simd_cycle  waves_per_simd  valu_share_of_wave_time  l2_hit_rate")
valu_active = {}
for k in sorted(cnt, key=lambda k: -dur.get(k, [0, 0])[1]):
    c = cnt[k]
    if not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_WAVE_CYCLES"):
        continue
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    valu_active[k] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
    print(f"{k:40s} {4.0 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / simd_cycles:8.3f} {4.0 * c['SQ_WAVE_CYCLES'] / simd_cycles:8.2f} "
          f"{c.get('SQ_ACTIVE_INST_VALU', 0.0) / c['SQ_WAVE_CYCLES']:8.3f} "
          f"{c.get('TCC_HIT_sum', 0.0) / max(c.get('TCC_REQ_sum', 0.0), 1.0):8.3f}")
# --traffic <out.json>: bytes the L2 requested from the fabric per launch of each kernel.  FETCH_SIZE / WRITE_SIZE are reported
# in KiB and are the L2's memory-side request counters (MI355X_MICROARCH.md, HBM): reads served by the Infinity Cache are
# INCLUDED, so this is L2-miss traffic, not HBM traffic.  FETCH_SIZE tallies every request at 64 bytes: measured against known
# bytes (profiles/r03_a_fetch_size_calibration_gather.txt) the factor is 0.96 for random 64-byte gathers and 1.89 for
# 128-byte ones (the guide's x2 is for wide streaming reads), so both ends are given: low = FETCH x1 + WRITE (a kernel
# of 64-byte gathers: the BVH4 walk's nodes), high = FETCH x2 + WRITE (every request a full 128-byte line).
if len(sys.argv) > 3 and sys.argv[2] == "--traffic":
    import json
    n_disp = defaultdict(int)
    for f in glob.glob(os.path.join(d, "tcc1", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                n_disp[short(r["Kernel_Name"])] += 1
    out = {}
    for k, n in n_disp.items():
        if not k.startswith("k_"):
            continue
        c = cnt[k]
        fetch = c.get("FETCH_SIZE", 0.0) * 1024.0
        write = c.get("WRITE_SIZE", 0.0) * 1024.0
        insts = c.get("SQ_INSTS_VALU", 0.0)
        out[k] = {"launches": n, "fetch_bytes_per_launch_raw": fetch / n, "write_bytes_per_launch": write / n,
                  "l2_fabric_bytes_per_launch_low": (fetch + write) / n, "l2_fabric_bytes_per_launch_high": (2.0 * fetch + write) / n,
                  "valu_active_per_simd_cycle": valu_active.get(k),
                  # wave-level VALU instructions per launch, and the share of the 64 lanes that were enabled when they ran
                  # (SQ_THREAD_CYCLES_VALU adds up the enabled lanes of every VALU instruction: k_jitter_table, 32 of 64
                  # lanes at work, reads 0.50)
                  "valu_instructions_per_launch": insts / n if insts else None,
                  "valu_lane_utilisation": (c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * insts)) if insts else None,
                  "l2_hit_rate": (c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"]) if c.get("TCC_REQ_sum") else None}
    json.dump({"source": os.path.basename(d.rstrip("/")), "kernels": out}, open(sys.argv[3], "w"), indent=1)
