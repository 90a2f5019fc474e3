#!/bin/bash
# kernel-trace stats of one bench frame without pass pipelining (clean per-kernel durations), for the current env
# usage: tools/ab_trace.sh <tag>   (env switches such as RT_HIP_NO_REFILL are inherited)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RT_HIP_PIPES=1
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-alone --steps 1 --warmup 0 > $OUT/log.txt 2>&1 || exit 1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cut -d, -f1-6 $f | cut -c1-140 | sed -n 1,8p
