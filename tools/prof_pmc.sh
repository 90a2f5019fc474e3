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
ARGS="$ROOT/bench.py --no-cpu-baseline --no-alone --steps 1 --warmup 0 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
# RT_PROF_PASSES="sq1 sq2 tcc1 tcc2 grbm" (optional): only these passes (scenes that take a minute to generate)
pass() {
  name=$1; shift
  if [ -n "${RT_PROF_PASSES:-}" ]; then case " $RT_PROF_PASSES " in *" $name "*) ;; *) return 0;; esac; fi
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || exit 1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass tcc1 FETCH_SIZE TCC_HIT_sum
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp2 TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass sq3 SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
echo done
