#!/bin/bash
# Developer tool (CPU): the host engines and the command line front end under AddressSanitizer +
# UndefinedBehaviorSanitizer.  Builds the emulated host library with the engine sources and the gpumt shim
# instrumented (the kernels and the fiber runtime stay as they are: custom stack switching confuses ASan),
# runs tests/test_emu_host_api.py and tests/test_emu_cli.py against it, then restores the normal library.
#   tools/emu_asan.sh            host engines + CLI
#   tools/emu_asan.sh kernels    the kernels themselves: every emulator kernel test with the kernels, the fiber
#                                runtime and the launch wrappers instrumented (heap accesses beyond a buffer and
#                                its documented slack are reported; the fibers' malloc'd stacks need no annotation)
set -e
if [ "$1" = kernels ]; then
  cd "$(dirname "$0")/../tests/emu"
  E=$PWD
  make -s libzmt_emu.so > /dev/null
  A=/tmp/zmt_kasan; mkdir -p $A
  SAN="-fsanitize=address -fno-omit-frame-pointer"
  rm -f $A/*.o
  for k in $(sed -n 's/^KERNELS := //p' Makefile); do
    g++ -O1 -g -std=c++17 -fPIC -DZMT_EMU $SAN -I. -I../../zstdmt_amd/csrc/hip -w -x c++ -c ../../zstdmt_amd/csrc/hip/$k.hip -o $A/$k.o &
  done
  wait
  for f in emu_runtime emu_api; do
    g++ -O1 -g -std=c++17 -fPIC -DZMT_EMU $SAN -I. -I../../zstdmt_amd/csrc/hip -w -c $f.cpp -o $A/$f.o
  done
  g++ -shared $SAN -Wl,-Bsymbolic -o $A/libzmt_emu.so $A/*.o
  cp libzmt_emu.so $A/normal.so
  trap 'cp $A/normal.so $E/libzmt_emu.so; touch $E/libzmt_emu.so' EXIT
  cp $A/libzmt_emu.so libzmt_emu.so; touch libzmt_emu.so
  cd ../..
  LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    python -m pytest ${ASAN_TESTS:-tests/test_emu_kernels.py tests/test_emu_zstd.py tests/test_emu_brotli.py tests/test_emu_snappy.py} \
      -q -x -p no:cacheprovider
  exit $?
fi
cd "$(dirname "$0")/../tests/emu"
E=$PWD
make -s libzstdmt_emu_host.so cli > /dev/null
A=/tmp/zmt_asan; mkdir -p $A
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
for f in lz4mt_engine zstdmt_engine brotlimt_engine snappymt_engine mt_pipe brotli_static; do
  gcc -O1 -g -fPIC -pthread $SAN -I../../include -I../../zstdmt_amd/csrc/host -Wa,-I../../zstdmt_amd/csrc/data \
      -c ../../zstdmt_amd/csrc/host/$f.c -o $A/host_$f.o
done
g++ -O1 -g -std=c++17 -fPIC -DZMT_EMU $SAN -I. -I../../zstdmt_amd/csrc/hip -w -c emu_gpumt.cpp -o $A/emu_gpumt.o
KOBJ=$(ls build/*.o | grep -v "/host_\|/emu_gpumt.o")
g++ -shared -pthread $SAN -Wl,-Bsymbolic -o $A/libzstdmt_emu_host.so $KOBJ $A/emu_gpumt.o $A/host_*.o
cp libzstdmt_emu_host.so $A/normal.so
trap 'cp $A/normal.so $E/libzstdmt_emu_host.so; touch $E/libzstdmt_emu_host.so' EXIT
cp $A/libzstdmt_emu_host.so libzstdmt_emu_host.so; touch libzstdmt_emu_host.so bin/*
cd ../..
LIBS="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
LD_PRELOAD="$LIBS" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_emu_host_api.py tests/test_emu_cli.py -q -x -p no:cacheprovider
