#!/bin/bash
# A/B builds of the wave-specialised forward kernel with parts switched off (FI_WS_DEBUG bit mask, conv_impl.h): which role
# bounds a stage.  bash tools/ws_variants.sh 6 22 21 19 17 25  ->  fedicra_amd/variants/libdbg<mask>.so
cd "$(dirname "$0")/../fedicra_amd/csrc" && mkdir -p ../variants build
for v in "$@"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DFI_WS_DEBUG=$v $EXTRA -c conv_bf16_v2.hip -o build/dbg$v.o & done; wait
for v in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libdbg$v.so $(ls build/*.o | grep -v "dbg\|conv_bf16_v2.o") build/dbg$v.o; done
ls ../variants
