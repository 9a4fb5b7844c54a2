#!/bin/bash
# Phase-ablation timing of k_rollout (profiling only): builds side libraries with -DEVAC_ABLATE=mask
# HERE (CPU container) into gpurun_out/ablate/, then `gpurun -- bash tools/ablate.sh run` times them.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
MASKS="0 1 2 4 8 16 3 6 7 15 31"
if [ "${1:-build}" = "build" ]; then
  mkdir -p "$ROOT/tools/ab_libs"
  for m in $MASKS; do
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -DEVAC_ABLATE=$m \
      "$ROOT/evacuation_amd/csrc/evac_api.hip" -o "$ROOT/tools/ab_libs/libevac_ablate_$m.so" &
  done
  wait
  ls "$ROOT/tools/ab_libs"
else
  for m in $MASKS; do
    EVAC_LIB="$ROOT/tools/ab_libs/libevac_ablate_$m.so" python3 "$ROOT/bench.py" --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask=%2d  us/step=%.3f  kernel_ms=%.4f' % ($m, d['ms_per_step']*1e3, d['roofline']['kernel_ms_per_launch']))"
  done
fi
