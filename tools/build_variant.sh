#!/bin/bash
# Build an experimental variant of the library: tools/build_variant.sh NAME FILE.hip [extra hipcc flags...]
# -> mad_amd/libmad_hip_NAME.so (the other objects are the regular build's)
set -e
cd "$(dirname "$0")/../mad_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8
obj=/tmp/variant_${name}_$(basename $src .hip).o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-inline-asm "$@" -c $src -o $obj
objs=""
for o in $(sed -n 's/^OBJS *:= *//p' Makefile); do
  if [ "$o" = "$(basename $src .hip).o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -o ../libmad_hip_${name}.so $objs
echo built mad_amd/libmad_hip_${name}.so
