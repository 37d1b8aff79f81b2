#!/bin/bash
# CPU-only sanitizer pass (GPU sanitizers are not available on the pool): the C oracle and the host side of
# the C-ABI library (table builder, fast-math validators, queue-encoding fuzz) under ASan + UBSan.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=${TMPDIR:-/tmp}
gcc -O1 -g -std=c11 -D_GNU_SOURCE -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined \
    -fno-sanitize-recover=undefined -shared -fPIC -o $TMP/libct_oracle_san.so $REPO/oracle/ct_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 $REPO/tests/sanitize_oracle_run.py $TMP/libct_oracle_san.so
if [ "$1" == "--host" ]; then
  cd $REPO/gymwipe_amd/csrc
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include \
      -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize=address -Xarch_host -fno-sanitize-recover=undefined \
      -shared -o $TMP/libgw_san.so -x hip $(grep '^SRCS' Makefile | sed 's/SRCS *:= *//')
  ASAN_OPTIONS=detect_leaks=0 GW_LIB=$TMP/libgw_san.so \
      LD_PRELOAD="$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)" python3 - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd() + "/../.."))
from gymwipe_amd import _native as nat
L = nat.lib()
for mult in (1, 3, 7, 15):
    assert L.gw_selftest_queue(123 + mult, 20000, mult, 65536) == 0 and L.gw_selftest_queue(5, 20000, mult, 40) == 0
for mult in (1, 3, 37, 100):                      # the generic kernel's run-length queues
    assert L.gw_selftest_runq(77 + mult, 20000, mult, 65536) == 0 and L.gw_selftest_runq(9, 20000, mult, 9) == 0
for D in (2, 4, 16, 32):
    cfg = nat.default_config(64, D); ns = C.c_int32()
    assert L.gw_selftest_fastmath(C.byref(cfg), C.byref(ns)) == 31
print("host side clean")
PY
fi
echo "sanitizers: clean"
