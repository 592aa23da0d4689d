#!/bin/bash
# The host side of libsert_hip.so under AddressSanitizer ("race detection / sanitizers": the reference has
# none; its NaN guards are sert_amd/models.py's per-batch finite check).  Builds sert_amd/variants/libsert_asan.so (host code
# instrumented, device code untouched: -fno-gpu-sanitize) and runs the given pytest selection against it:
#     tools/run_asan.sh -m "not gpu" -k "row_exchange or index or capi"        (here: the host-only entry points)
# (On a GPU box the HSA runtime aborts inside hipGetDeviceCount when the sanitizer runtime is preloaded -- round 5,
#  gpurun_out/r05e/asan_gpu.txt -- so this covers the host-only entry points: the per-batch inverted index, the row-exchange
#  lists, symbol / struct checks.  Device-side memory errors are what the parity tests and rocgdb are for.)
# Leak detection is off (the Python interpreter and the HIP runtime keep their allocations until exit).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/sert_amd/variants/libsert_asan.so
if [ ! -f $LIB ] || [ -n "$(find $ROOT/sert_amd/csrc $ROOT/include -newer $LIB -name '*.h*' | head -1)" ]; then
  bash $ROOT/tools/build_variant.sh asan -fsanitize=address -fno-gpu-sanitize -g -O1 -shared-libsan > /dev/null
fi
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $ROOT
SERT_LIB=$LIB LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 python -m pytest tests -q -x "$@"
