#!/bin/bash
# scripts/build_variant.sh <name> [extra hipcc flags...]: development build of the library with extra
# flags (e.g. -DCAMD_PROFILE) into build/libconcrete_amd_<name>.so; select it with CONCRETE_AMD_LIB=<path>.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build
cd $ROOT/concrete_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" -x hip \
  abi.hip keyswitch.hip keyswitch_mfma.hip linear.hip pbs_generic.hip pbs_block.hip pbs_wave.hip pbs_wave512.hip probe.hip client.cpp \
  -o $ROOT/build/libconcrete_amd_$NAME.so -lpthread
echo built $ROOT/build/libconcrete_amd_$NAME.so
