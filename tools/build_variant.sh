#!/bin/bash
# Build an experimental variant of libsert_hip.so:  tools/build_variant.sh <name> [-DFOO=1 ...]
# -> sert_amd/variants/libsert_<name>.so ; run with SERT_LIB=<that path>
# The opt-in GEMM variants of csrc/variants/ (strip / role-specialised / 256x256-tile kernels that lost
# their A/B) are compiled in with:  tools/build_variant.sh variants -DSERT_VARIANTS
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/sert_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$ROOT/include "$@" \
    $ROOT/sert_amd/csrc/sert_hip.hip -o $ROOT/sert_amd/variants/libsert_$NAME.so -ldl
echo $ROOT/sert_amd/variants/libsert_$NAME.so
