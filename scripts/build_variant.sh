#!/bin/bash
# scripts/build_variant.sh <name> [extra hipcc flags...]: development build of the library with extra
# flags (e.g. -DCAMD_PROFILE) into build/libconcrete_amd_<name>.so; select it with CONCRETE_AMD_LIB=<path>.
# VARIANT_SOURCES (default: every source) names the sources that are recompiled with the flags; the
# other objects are taken from the regular build (concrete_amd/_lib/obj, python __graft_entry__.py).
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
ALL="abi.hip keyswitch.hip keyswitch_mfma.hip linear.hip pbs_generic.hip pbs_block.hip pbs_wave.hip pbs_wave8.hip pbs_wave16k.hip pbs_wave512.hip probe.hip client.cpp"
SRCS=${VARIANT_SOURCES:-$ALL}
mkdir -p $ROOT/build/obj_$NAME
cd $ROOT/concrete_amd/csrc
OBJS=""
for f in $ALL; do
  if [[ " $SRCS " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -x hip -c $f -o $ROOT/build/obj_$NAME/$f.o &
    OBJS="$OBJS $ROOT/build/obj_$NAME/$f.o"
  else
    OBJS="$OBJS $ROOT/concrete_amd/_lib/obj/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/build/libconcrete_amd_$NAME.so -lpthread
echo built $ROOT/build/libconcrete_amd_$NAME.so
