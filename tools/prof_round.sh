#!/bin/bash
# Round profile set on the GPU box: kernel-trace stats of the default run and of one pass in flight, then the PMC passes
# (first-bounce traversal isolated with RT_HIP_TAIL_DEPTH=2).  usage: tools/prof_round.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for mode in default pipes1; do
  OUT=$ROOT/gpurun_out/stats_${TAG}_$mode
  mkdir -p $OUT
  if [ $mode = pipes1 ]; then export RT_HIP_PIPES=1; else unset RT_HIP_PIPES; fi
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-alone --steps 2 --warmup 1 "$@" > $OUT/trace.log 2>&1 || { echo "stats $mode failed"; tail -3 $OUT/trace.log; exit 1; }
  echo "stats $mode done"
done
unset RT_HIP_PIPES
cd $ROOT
RT_HIP_TAIL_DEPTH=2 tools/prof_pmc.sh $TAG "$@" || exit 1
python3 tools/pmc_summary.py gpurun_out/prof_$TAG --traffic gpurun_out/prof_$TAG/traffic.json > gpurun_out/prof_$TAG/summary.txt
echo "pmc done"
