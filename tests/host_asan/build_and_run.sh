#!/bin/bash
# Builds the HOST half of libmod16hip with AddressSanitizer + UndefinedBehaviorSanitizer against the
# HIP runtime stand-in of this directory and runs the driver:  bash tests/host_asan/build_and_run.sh [OUTDIR]
# (hipcc --cuda-host-only: the library's own source, its host side exactly as shipped; no GPU needed)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${1:-$(mktemp -d)}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
# the driver and the stub are small: compile them while the library compiles
$HIPCC --cuda-host-only $SAN -std=c++17 -w -c "$ROOT/mod16_amd/csrc/mod16_capi.hip" -o "$OUT/capi.o" &
$HIPCC --cuda-host-only $SAN -std=c++17 -w -c "$HERE/hip_stub.hip" -o "$OUT/stub.o"
$HIPCC --cuda-host-only $SAN -std=c++17 -w -x hip -c "$HERE/driver.cpp" -o "$OUT/driver.o"
wait
# host-only objects refer to the device code object of their translation unit by a hashed symbol: define them
: > "$OUT/fatbin.c"
for o in capi stub driver; do
  for s in $(nm "$OUT/$o.o" | awk '$1 == "U" && $2 ~ /^__hip_fatbin/ {print $2}'); do echo "const char $s[8] = {0};" >> "$OUT/fatbin.c"; done
done
/opt/rocm/lib/llvm/bin/clang $SAN -c "$OUT/fatbin.c" -o "$OUT/fatbin.o"
/opt/rocm/lib/llvm/bin/clang++ $SAN "$OUT/driver.o" "$OUT/capi.o" "$OUT/stub.o" "$OUT/fatbin.o" -lpthread -o "$OUT/host_asan"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 MOD16_HOST_THREADS=3 "$OUT/host_asan"
# ... and the harness sees a fault when there is one (exit code != 0 expected)
if "$OUT/host_asan" --fault > "$OUT/fault.log" 2>&1; then echo "host_asan: the planted fault was NOT detected"; cat "$OUT/fault.log"; exit 1; fi
grep -q "lies outside every live device allocation" "$OUT/fault.log" && echo "host_asan: planted fault detected: $(grep 'hip_stub:' "$OUT/fault.log" | head -1 | cut -c1-160)"
