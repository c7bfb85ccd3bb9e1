#!/bin/bash
# Developer tool: build libzstdmt_amd.so variants with one HIP source recompiled under extra -D flags
# (A/B measurements on the GPU box: ZMT_LIB=zstdmt_amd/lib/variants/<name>.so python tools/dec_prof.py ...).
#   tools/variant_build.sh <source.hip> <name> "<flags>" [<name> "<flags>" ...]
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
OBJ=zstdmt_amd/build
OUT=zstdmt_amd/lib/variants
mkdir -p $OUT
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS \
      -c zstdmt_amd/csrc/hip/$SRC -o $OBJ/variant_$NAME.o
  OBJS=$(ls $OBJ/*.o | grep -v "/variant_" | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OBJS $OBJ/variant_$NAME.o -lpthread
  echo "built $OUT/$NAME.so ($FLAGS)"
done
