#!/bin/bash
# A/B builds of conv_fwd_ws2_kernel with parts switched off (FI_WS2_DEBUG bit mask, conv_ws2.h): which role bounds a stage.
#   bash tools/ws2_variants.sh 0 22 6 16 32 17 8  ->  variants/libws2dbg<mask>.so   (build container; the .so files travel with gpurun)
# EXTRA="-DFOO" adds compiler flags; NAME=suffix renames the outputs (libws2dbg<mask><suffix>.so)
cd "$(dirname "$0")/../fedicra_amd/csrc" && mkdir -p ../../variants build
for v in "$@"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DFI_WS2_DEBUG=$v $EXTRA -c conv_bf16_ws2.hip -o build/ws2dbg$v$NAME.o & done; wait
for v in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libws2dbg$v$NAME.so $(ls build/*.o | grep -v "dbg\|conv_bf16_ws2.o") build/ws2dbg$v$NAME.o; done
ls ../../variants
