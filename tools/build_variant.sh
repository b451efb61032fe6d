#!/bin/bash
# Debug / experiment builds of libp3v.so that differ in ONE translation unit: tools/build_variant.sh <name> <file.hip> [-D...]
# -> build/libp3v_<name>.so (objects of the other sources are cached under build/obj).  Never the shipped library.
set -e
cd "$(dirname "$0")/../phi-3-vision-mlx_amd/csrc"
NAME=$1; UNIT=$2; shift 2
OBJ=../../build/obj; mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
for f in *.hip; do
  [ "$f" = "$UNIT" ] && continue
  if [ ! -f $OBJ/${f%.hip}.o ] || [ $f -nt $OBJ/${f%.hip}.o ] || [ p3v_common.h -nt $OBJ/${f%.hip}.o ] || [ ../../include/p3v.h -nt $OBJ/${f%.hip}.o ]; then
    hipcc $FLAGS -c $f -o $OBJ/${f%.hip}.o &
  fi
done
hipcc $FLAGS "$@" -c $UNIT -o $OBJ/${UNIT%.hip}_$NAME.o &
wait
OBJS=$(for f in *.hip; do [ "$f" = "$UNIT" ] && echo $OBJ/${UNIT%.hip}_$NAME.o || echo $OBJ/${f%.hip}.o; done)
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs $OBJS -o ../../build/libp3v_$NAME.so
echo "built build/libp3v_$NAME.so"
