#!/bin/bash
# Fast variant build of libfedicra_hip.so for conv_fwd_dma_kernel A/B runs: only conv_api + the two conv_dma units are recompiled with
# $EXTRA (e.g. -DFI_TRACE, -DFI_DMA_DEBUG=1), the other objects come from fedicra_amd/csrc/build (run make first).  -> variants/<name>.so
set -e
NAME=$1; ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/fedicra_amd/csrc; T=$(mktemp -d /tmp/dmav_XXXX)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $EXTRA"
for u in conv_api conv_bf16_dma conv_f16_dma; do (cd $C && hipcc $FL -c $u.hip -o $T/$u.o) & done; wait
mkdir -p $ROOT/variants
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/$NAME.so $(ls $C/build/*.o | grep -v "conv_api.o\|conv_bf16_dma.o\|conv_f16_dma.o") $T/conv_api.o $T/conv_bf16_dma.o $T/conv_f16_dma.o
rm -rf $T; echo built variants/$NAME.so
