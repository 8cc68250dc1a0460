#!/bin/bash
# Variant build of libfedicra_hip.so for kernel A/B runs: the listed units are recompiled with $EXTRA (e.g. -DFI_ROWS3D_PIPE=0), the
# other objects come from fedicra_amd/csrc/build (run make first).  bash tools/build_variant.sh <name> <unit> [unit ...] -> variants/<name>.so
set -e
NAME=$1; shift; ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/fedicra_amd/csrc; T=$(mktemp -d /tmp/var_XXXX)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $EXTRA"
PAT=""
for u in "$@"; do (cd $C && hipcc $FL -c $u.hip -o $T/$u.o) & PAT="$PAT\|/$u.o"; done; wait
mkdir -p $ROOT/variants
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/$NAME.so $(ls $C/build/*.o | grep -v "${PAT:2}") $T/*.o
rm -rf $T; echo built variants/$NAME.so
