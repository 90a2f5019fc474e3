#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE (the L2's fabric-side read requests, Infinity-Cache hits included:
# MI355X_MICROARCH.md §HBM) for the access pattern of the traversal kernels — every lane gathers its own 64-byte (or
# 128-byte) record at a random place of a table far larger than the L2s — against KNOWN bytes:
#   lane fetches x bytes per fetch x L2 miss share (TCC_MISS_sum / TCC_REQ_sum of the same kernel).
# The guide's x2 was measured on wide streaming reads; this gives the factor for gathers.
# usage (GPU box): tools/calib_fetch.sh <tag>      -> gpurun_out/calib_<tag>/summary.txt
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$ROOT/tools/micro/gather_bench
[ -x $BIN ] || { echo "build tools/micro/gather_bench first (hipcc --offload-arch=gfx950 -O3)"; exit 1; }
cd /tmp
for B in 64 128; do
  for MIB in 1024; do
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_${B}_$MIB -- $BIN calib $B $MIB > $OUT/f_${B}_$MIB.log 2>&1 || exit 1
    timeout -s KILL 120 rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $OUT/m_${B}_$MIB -- $BIN calib $B $MIB > $OUT/m_${B}_$MIB.log 2>&1 || exit 1
  done
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, os, re, sys
d = sys.argv[1]
print("FETCH_SIZE calibration for random per-lane gathers (table 1024 MiB, 16 waves per CU, 64 lanes)")
for B in (64, 128):
    tot = {}
    for kind in ("f", "m"):
        for f in glob.glob(os.path.join(d, f"{kind}_{B}_1024", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "chase" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    log = open(os.path.join(d, f"f_{B}_1024.log")).read()
    fetches = float(re.search(r"lane_fetches_all_launches (\d+)", log).group(1))
    miss = tot.get("TCC_MISS_sum", 0.0) / max(tot.get("TCC_REQ_sum", 0.0), 1.0)
    raw = tot.get("FETCH_SIZE", 0.0) * 1024.0
    known = fetches * B * miss
    print(f"bytes/fetch {B:3d}: lane fetches {fetches:.4g}, TCC_REQ {tot.get('TCC_REQ_sum', 0):.4g}, L2 miss share {miss:.3f}, "
          f"known fabric bytes {known:.4g}, FETCH_SIZE raw {raw:.4g} -> factor {known / max(raw, 1.0):.3f}")
PY
cat $OUT/summary.txt
