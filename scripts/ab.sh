#!/bin/bash
# A/B timing of library builds inside one GPU call (clocks differ between boxes, so only numbers
# from the same call compare).  Here:  bash scripts/ab.sh save NAME  copies the built libmrt.so
# to raytracing_amd/variants/libmrt_NAME.so (git-ignored).  On the GPU:  bash scripts/ab.sh run
# times every saved variant ROUNDS times, interleaved.
set -u
cd "$(dirname "$0")/.."
OUT=raytracing_amd/variants
if [ "${1:-}" = save ]; then
  mkdir -p $OUT && cp raytracing_amd/libmrt.so $OUT/libmrt_$2.so && echo "saved $2"
elif [ "${1:-}" = clear ]; then
  rm -f $OUT/libmrt_*.so
else
  for r in $(seq ${ROUNDS:-3}); do
    for f in $OUT/libmrt_*.so; do
      n=$(basename $f .so); n=${n#libmrt_}
      MRT_LIB=$f timeout -k 10 100 python3 -u scripts/${PROBE:-perf_probe.py} ${PROBE_ARGS:-} 2>&1 | grep -v amdgpu.ids | grep -E "${PROBE_GREP:-.}" | sed "s/^/$n  /" || exit 1
    done
  done
fi
