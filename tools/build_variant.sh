#!/bin/bash
# Developer A/B: the product library with ONE source file rebuilt under extra -D flags.
#   bash tools/build_variant.sh <name> <file.hip> "<flags>"   ->  bayeformers_amd/lib/libbayeformers_amd_<name>.so
# Use with BF_LIB_PATH=$PWD/bayeformers_amd/lib/libbayeformers_amd_<name>.so (python -m bayeformers_amd.build first).
set -e
NAME=$1; FILE=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/bayeformers_amd/csrc/_obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $ROOT/bayeformers_amd/csrc/$FILE -o $OBJ/${FILE%.hip}.$NAME.o
OBJS=""
for o in $OBJ/*.o; do
    case $o in *.dev.o) continue;; esac
    b=$(basename $o .o)
    case $b in *.*) continue;; esac            # other variants
    if [ "$b" = "${FILE%.hip}" ]; then OBJS="$OBJS $OBJ/${FILE%.hip}.$NAME.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/bayeformers_amd/lib/libbayeformers_amd_$NAME.so $OBJS
echo $ROOT/bayeformers_amd/lib/libbayeformers_amd_$NAME.so
