#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of libevac (SURVEY.md 5 "sanitizers": ASan on the host library; GPU
# ASan / xnack+ code objects are not available on the pool) and a run of tools/asan/host_driver.c against it.
#   bash tools/asan_host.sh        -> prints "asan host driver: ok", exit 0; any ASan/UBSan report fails the run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/evac_asan
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O1 -g --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize \
  -fno-sanitize-recover=undefined "$ROOT/evacuation_amd/csrc/evac_api.hip" -o "$OUT/libevac_asan.so"
RT=$(dirname "$($HIPCC --print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || echo /opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so)")
[ -f "$RT/libclang_rt.asan-x86_64.so" ] || RT=$(dirname "$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address,undefined -shared-libsan -I"$ROOT/include" "$ROOT/tools/asan/host_driver.c" \
  -L"$OUT" -levac_asan -Wl,-rpath,"$OUT" -Wl,-rpath,"$RT" -o "$OUT/host_driver"
ASAN_OPTIONS=detect_leaks=1:halt_on_error=1:abort_on_error=0 LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" "$OUT/host_driver"
