#!/bin/bash
# Interleaved A/B of library builds on one GPU box: tools/ab_bench.sh "<bench args>" lib1.so lib2.so ...   (rounds: $ROUNDS, default 3)
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${ROUNDS:-3}); do
  for lib in "$@"; do
    EVAC_LIB="$ROOT/$lib" python3 "$ROOT/bench.py" --no-cpu-baseline --no-step-api $ARGS 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s value=%.4g us/step=%.4f kernel_ms=%.4f dense_ms=%.4f' % ('$lib', d['value'], d['ms_per_step']*1e3, r['kernel_ms_per_launch'], r['kernel_ms_per_launch_dense']))"
  done
done
