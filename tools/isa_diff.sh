#!/bin/bash
# Developer tool (CPU, hipcc cross-compiles): compare the gfx950 code hipcc generates for one kernel source
# at a git revision and in the working tree -- instruction lines only, labels / directives / comments dropped.
# "0 differing lines" = the change cannot alter anything measured on the device.
#   tools/isa_diff.sh lz4_enc3.hip [rev=HEAD] [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
SRC=$1; REV=${2:-HEAD}; shift; shift || true
T=$(mktemp -d)
git show $REV:zstdmt_amd/csrc/hip/$SRC > $T/old.hip
strip() { grep -v '^\s*;\|^\.L\|^\s*\.\|^$\|^__hip_cuid' $1 | sed 's/;.*//'; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izstdmt_amd/csrc/hip --cuda-device-only -S -x hip $T/old.hip -o $T/old.s "$@" 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izstdmt_amd/csrc/hip --cuda-device-only -S zstdmt_amd/csrc/hip/$SRC -o $T/new.s "$@" 2>/dev/null
strip $T/old.s > $T/old.i; strip $T/new.s > $T/new.i
echo "$SRC: $(wc -l < $T/old.i) instruction lines at $REV, $(wc -l < $T/new.i) in the working tree, $(diff $T/old.i $T/new.i | grep -c '^[<>]') differing lines"
rm -rf $T
