#!/bin/bash
# Cross-compile an experiment variant of the library here (no GPU needed) into build_ab/libkgan_<tag>.so; the file travels
# to the GPU box with the tree (git-ignored *.so) and is selected there with KG_LIB=build_ab/libkgan_<tag>.so.
#   tools/build_variant.sh <tag> "<extra hipcc flags>" [source.hip ...]
# Objects of unchanged sources are cached in /tmp/kgobj/base (rebuilt when a source is newer); only the sources named on
# the command line (default: kg_conv.hip) are compiled with the extra flags.
set -eu
TAG=$1; FLAGS=${2:-}; shift; shift || true
VAR_SRCS=${*:-kg_conv.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/kinetic-gan_amd/csrc
BASE=/tmp/kgobj/base; VAR=/tmp/kgobj/$TAG
mkdir -p $BASE $VAR $ROOT/build_ab
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I $ROOT/include -I $CS"
OBJS=""
for f in $CS/*.hip; do
    b=$(basename $f .hip)
    if echo " $VAR_SRCS " | grep -q " $b.hip "; then
        $CC $FLAGS -c $f -o $VAR/$b.o &
        OBJS="$OBJS $VAR/$b.o"
    else
        if [ ! -f $BASE/$b.o ] || [ $f -nt $BASE/$b.o ] || [ $CS/kg_common.h -nt $BASE/$b.o ] || [ $ROOT/include/kgan_hip.h -nt $BASE/$b.o ]; then
            $CC -c $f -o $BASE/$b.o &
        fi
        OBJS="$OBJS $BASE/$b.o"
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/libkgan_$TAG.so $OBJS -ldl
echo built $ROOT/build_ab/libkgan_$TAG.so
