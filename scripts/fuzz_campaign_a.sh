# The parity campaigns on one library state (through gpurun): random poses and soups through both shaders against the
# oracle, random API sequences against fresh contexts, rasterizer frames against the oracle's scan-line rasterizer.
# Logs under gpurun_out/r3fuzz/.  usage: bash scripts/fuzz_campaign_a.sh [poses] [soups] [api ops] [seed]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3fuzz
P=${1:-2500}; S=${2:-300}; A=${3:-2000}; SEED=${4:-31}
timeout -k 10 1100 python3 -u scripts/parity_fuzz.py $P $SEED > gpurun_out/r3fuzz/poses.log 2>&1 || { echo "pose campaign FAILED"; tail -5 gpurun_out/r3fuzz/poses.log; exit 1; }
tail -1 gpurun_out/r3fuzz/poses.log
timeout -k 10 1100 python3 -u scripts/parity_fuzz.py --soups $S $((SEED+1)) > gpurun_out/r3fuzz/soups.log 2>&1 || { echo "soup campaign FAILED"; tail -5 gpurun_out/r3fuzz/soups.log; exit 1; }
tail -1 gpurun_out/r3fuzz/soups.log
timeout -k 10 1100 python3 -u scripts/api_fuzz.py $A $((SEED+2)) > gpurun_out/r3fuzz/api.log 2>&1 || { echo "api campaign FAILED"; tail -5 gpurun_out/r3fuzz/api.log; exit 1; }
tail -1 gpurun_out/r3fuzz/api.log
timeout -k 10 1100 python3 -u scripts/raster_fuzz.py 400 30 $((SEED+3)) > gpurun_out/r3fuzz/raster.log 2>&1 || { echo "raster campaign FAILED"; tail -5 gpurun_out/r3fuzz/raster.log; exit 1; }
tail -1 gpurun_out/r3fuzz/raster.log
