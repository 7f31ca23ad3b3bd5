#!/bin/bash
# rocprofv3 evidence of one round, collected on the GPU box.  Usage (repo root, through gpurun):
#   bash profiles/run_profile_r02.sh <tag>
# Writes gpurun_out/prof_<tag>/: one kernel-trace + stats run over the whole benchmark, then separate --pmc passes
# (never combined with tracing) per configuration over scripts/config_probe.py.  profiles/summarize_r02.py condenses
# them into profiles/<tag>/.
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/trace_stdout.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace_stdout.log; exit 1; }
echo "trace done"
# the headline alone (no sub-benchmarks): the average duration of its kernel in this trace is the one to compare with
# bench.py's roofline.kernel_ms
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu --no-extra > $OUT/trace_headline_stdout.log 2>&1 || { echo "headline trace run failed"; tail -5 $OUT/trace_headline_stdout.log; exit 1; }
echo "headline trace done"
[ "${2:-}" = "trace-only" ] && exit 0
pass() {  # config, pass name, counters...
  local cfg=$1 name=$2; shift 2
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${cfg}_$name -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/config_probe.py $cfg 3 > $OUT/pmc_${cfg}_${name}_stdout.log 2>&1 || { echo "pmc pass $cfg $name failed"; tail -3 $OUT/pmc_${cfg}_${name}_stdout.log; return 1; }
  echo "pass $cfg $name done"
}
for cfg in c2 c2r c2p c4 c3 c5; do
  pass $cfg sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD && \
  pass $cfg sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU && \
  pass $cfg fetch FETCH_SIZE && \
  pass $cfg write WRITE_SIZE && \
  pass $cfg grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
done
pass c2 sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH
pass c3 tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
echo profile done; ls $OUT
