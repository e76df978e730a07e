#!/bin/bash
# CPU only (build container): the HOST side of the C ABI — capi.hip's argument checks, size queries, validators — under
# AddressSanitizer.  capi.hip is recompiled with -fsanitize=address for the host pass only (-fno-gpu-sanitize: device code as
# shipped; GPU sanitizers are not available on the pool and are never used), linked with the shipped objects of the other
# files, and tests/test_capi_argument_checks.py + the validator tests of tests/test_host_logic.py run against it with the
# clang ASAN runtime preloaded.  Any out-of-bounds access of a host array in those paths aborts the run.
#   tools/asan_host.sh            -> "N passed" and exit code 0, or ASAN's report
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/active-3d-vision-and-touch_amd"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p "$PKG/build/asan"
$HIPCC -O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -c "$PKG/csrc/capi.hip" -o "$PKG/build/asan/capi.o"
OBJS=$(ls "$PKG"/build/*.o | grep -v "/capi.o\|-hip-amdgcn" | tr '\n' ' ')
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address "$PKG/build/asan/capi.o" $OBJS -o "$PKG/build/asan/liba3vt_asan.so"
RT=$($HIPCC -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 LD_PRELOAD="$RT" A3VT_LIB="$PKG/build/asan/liba3vt_asan.so" \
  python -m pytest tests/test_capi_argument_checks.py tests/test_host_logic.py -x -q \
  -k "argument or counters or validators or size_queries or refuse or header_symbols or split or csr_rows"
