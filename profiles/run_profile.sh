#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box.  Usage (from the repo root, via gpurun):
#   bash profiles/run_profile.sh <tag> [extra bench flags]
# Writes gpurun_out/prof_<tag>/: a kernel-trace + stats run and separate --pmc passes (never
# combined with tracing).  Copy the summaries you want to keep into profiles/.
set -u
TAG=${1:-r01}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_stdout.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace_stdout.log; exit 1; }
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH --no-extra > $OUT/pmc_${name}_stdout.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $OUT/pmc_${name}_stdout.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD && \
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo profile done; ls $OUT
