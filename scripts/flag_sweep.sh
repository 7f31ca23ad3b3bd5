#!/bin/bash
# Builds libmrt variants with extra compiler flags for kernels.hip (CPU side, here) so that one GPU call
# can time them all:  bash scripts/flag_sweep.sh build ; then on the GPU:  bash scripts/flag_sweep.sh run
# A tuning aid; variants are written to raytracing_amd/variants/ (git-ignored).
set -u
cd "$(dirname "$0")/.."
CS=raytracing_amd/csrc
OUT=raytracing_amd/variants
BASE="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -w"
declare -A V
V[fwd]=""
V[rev]="-DMRT_REVERSE_TILES"
if [ "${1:-}" = build ]; then
  mkdir -p $OUT
  make -C $CS -j8 >/dev/null
  for n in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc $BASE ${V[$n]} -c $CS/kernels.hip -o $OUT/kernels_$n.o 2>$OUT/$n.log &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libmrt_$n.so $OUT/kernels_$n.o $CS/build/raster.o $CS/build/lbvh.o $CS/build/mrt_api.o $CS/build/bvh_builder.o $CS/build/camera.o &&
      echo "built $n" || echo "FAILED $n: $(tail -2 $OUT/$n.log)" ) &
  done
  wait
else
  for f in $OUT/libmrt_*.so; do
    n=$(basename $f .so); n=${n#libmrt_}
    echo "== $n"
    MRT_LIB=$f timeout -k 10 100 python3 -u scripts/${PROBE:-perf_probe.py} ${PROBE_ARGS:-} 2>&1 | grep -v amdgpu.ids | grep -E "${PROBE_GREP:-.}" || exit 1
  done
fi
