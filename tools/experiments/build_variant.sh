#!/bin/bash
# tools/build_variant.sh <tag> <file.hip> <flags...>: rebuild ONE translation unit with extra flags and link it with the cached objects
# of the regular build into ratrack_amd/lib/variants/librtk_<tag>.so (select it with RTK_SO_PATH=...; experiments only).
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
V=ratrack_amd/lib/variants; mkdir -p $V
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-result "$@" \
  -I include -I ratrack_amd/csrc -c $src -o $V/${base}_$tag.o
objs=$(ls ratrack_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/librtk_$tag.so $objs $V/${base}_$tag.o
rm -f $V/${base}_$tag.o
echo $V/librtk_$tag.so
