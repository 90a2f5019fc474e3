#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 kernel trace (tools/ktrace.sh output dir): per kernel start / end / queue, and how
many rt kernels were in flight over time.  usage: tools/timeline.py gpurun_out/kt_<tag> [--rows]"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
short = lambda n: n.split("(")[0].replace("void ", "").replace("rt::", "")[:40]
rows = [r for r in rows if short(r["Kernel_Name"]).startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
if "--rows" in sys.argv:
    for r in rows:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        print(f"{s:9.3f} {e:9.3f} {e - s:8.3f} q={r['Queue_Id']} {short(r['Kernel_Name'])}")
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
hist, cur, last = {}, 0, ev[0][0]
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += dlt
    last = t
tot = sum(hist.values())
print("kernels in flight: share of the traced span " + ", ".join(f"{k}: {v / tot:.3f}" for k, v in sorted(hist.items())))
