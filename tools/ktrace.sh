#!/bin/bash
# rocprofv3 kernel trace of one bench frame -> per-kernel summary (calls, total ms, avg us).
# usage (on the GPU box): tools/ktrace.sh <tag> [bench args...]; environment switches are inherited.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-alone --steps 1 --warmup 0 "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
dur = defaultdict(lambda: [0, 0])
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
        dur[k][0] += 1
        dur[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(dur.items(), key=lambda x: -x[1][1]):
    print(f"{k:52s} {c:6d} {t / 1e6:10.3f} ms {t / c / 1e3:10.1f} us")
PY
tail -1 $OUT/trace.log | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('value=%.1f ms/step=%.2f' % (j['value'], j['ms_per_step']))"
