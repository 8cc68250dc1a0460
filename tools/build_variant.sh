#!/bin/bash
# Build libfedicra_hip.so from the csrc/ of a given commit (or "WORK" = working tree) into variants/<name>.so
set -e
REV=$1; NAME=$2; ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/var_XXXX)
if [ "$REV" = "WORK" ]; then
  mkdir -p $T/fedicra_amd && cp -r $ROOT/fedicra_amd/csrc $T/fedicra_amd/ && cp -r $ROOT/include $T/ && rm -rf $T/fedicra_amd/csrc/build
else
  git -C $ROOT archive $REV fedicra_amd/csrc include | tar -x -C $T
fi
make -C $T/fedicra_amd/csrc -j8 EXTRA="$EXTRA" > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
mkdir -p $ROOT/variants && cp $T/fedicra_amd/libfedicra_hip.so $ROOT/variants/$NAME.so && rm -rf $T
echo built variants/$NAME.so
