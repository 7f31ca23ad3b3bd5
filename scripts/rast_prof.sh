# Rasterizer tuning aid (through gpurun): per-kernel times of the 1,000,000-triangle 4K frame.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 scripts/config_probe.py c5 10
mkdir -p gpurun_out/r3rast
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3rast -o c5 -- python3 scripts/config_probe.py c5 10 > gpurun_out/r3rast/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3rast/**/c5_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'raster' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
