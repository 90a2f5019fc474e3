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
# --traffic <out.json>: HBM-side bytes per launch of each kernel, FETCH_SIZE / WRITE_SIZE are reported in KiB;
# FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B; calibrated
# there for wide streaming reads only, so for these gather kernels it is an upper estimate of the correction).
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
        fetch = cnt[k].get("FETCH_SIZE", 0.0) * 1024.0
        write = cnt[k].get("WRITE_SIZE", 0.0) * 1024.0
        out[k] = {"launches": n, "fetch_bytes_per_launch_raw": fetch / n, "write_bytes_per_launch": write / n,
                  "traffic_bytes_per_launch": (2.0 * fetch + write) / n,
                  "valu_active_per_simd_cycle": valu_active.get(k)}
    json.dump({"source": os.path.basename(d.rstrip("/")), "kernels": out}, open(sys.argv[3], "w"), indent=1)
