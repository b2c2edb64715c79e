#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (GPU sanitizers are not available on this pool):
#   * the host twin of the device templates (tests/hosttwin: the field, group-law, window, table and lane-workspace code that the
#     kernels instantiate), under the whole tests/test_hosttwin_*.py suite;
#   * the C oracle (oracle/ecoracle.c), under tests/test_oracle_c.py and tests/test_oracle_ecdsa.py.
# Usage: tools/hosttwin_sanitize.sh [outfile]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/dev/stdout}
B=$(mktemp -d /tmp/ecgpu_san.XXXXXX)
CXX=/opt/rocm/lib/llvm/bin/clang++
CC=/opt/rocm/lib/llvm/bin/clang
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan"
$CXX -O1 -g -std=c++17 -fPIC -shared $SAN -Irustcrypto-elliptic-curves_amd/csrc -Wall -Wno-unused-function tests/hosttwin/hosttwin_*.cpp -o $B/libechosttwin.so
$CC -O1 -g -fPIC -shared -pthread $SAN -Wall -Wno-unused-function oracle/ecoracle.c -o $B/libecoracle.so
{
  echo "# clang $($CXX --version | head -1)"
  echo "# flags: $SAN (undefined behaviour aborts the run; leak checking off: the interpreter's own allocations)"
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 ECGPU_HOSTTWIN_LIB=$B/libechosttwin.so ECGPU_ORACLE_LIB=$B/libecoracle.so \
    python -m pytest tests/test_hosttwin_field.py tests/test_hosttwin_jacobian.py tests/test_hosttwin_k256.py tests/test_hosttwin_k256_fast.py tests/test_hosttwin_nist.py \
      tests/test_hosttwin_straus.py tests/test_hosttwin_vbct.py tests/test_hosttwin_ct_and_vb.py tests/test_oracle_c.py tests/test_oracle_ecdsa.py -q -p no:cacheprovider 2>&1 | tail -15
} > "$OUT"
rm -rf $B
