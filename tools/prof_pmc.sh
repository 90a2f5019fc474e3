#!/bin/bash
# Collects rocprofv3 summaries on the GPU box: one kernel-trace/stats run, then PMC passes, each its
# own run (no trace domains mixed with --pmc; <= 8 SQ / 4 TCC counters per pass).
# usage: tools/prof_pmc.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || exit 1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass tcc1 FETCH_SIZE TCC_HIT_sum
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo done
