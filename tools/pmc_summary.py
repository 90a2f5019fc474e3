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
