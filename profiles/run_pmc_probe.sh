#!/bin/bash
# PMC passes over one perf_probe configuration.  Usage: bash profiles/run_pmc_probe.sh <tag> <probe args...>
# (PROBE=c3_probe.py PROBE_FIXED=' ' selects another script under scripts/)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/scripts/${PROBE:-perf_probe.py} ${PROBE_FIXED:---iters 5} $*"
pass() { local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > $OUT/${name}_stdout.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/${name}_stdout.log; return 1; }; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD && \
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU && \
pass sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM && \
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum && \
pass fetch FETCH_SIZE && \
pass grbm GRBM_GUI_ACTIVE
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$OUT/*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(d)):
        if 'render_tiles' not in r['Kernel_Name'] and 'raster' not in r['Kernel_Name']: continue
        k = (r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        print("%-42s %-24s %.5g  (n=%d)" % (k[0], k[1], v[0] / v[1], v[1]))
PY
