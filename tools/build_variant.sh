#!/bin/bash
# Build an experimental variant of the library (timing-only / A-B builds): conv_igemm.hip recompiled with extra flags,
# linked with the regular objects of the other sources -> weaklysuperviseddl_amd/csrc/exp/libwsdl_<name>.so
# usage: tools/build_variant.sh <name> [-DFLAG ...]        then run with WSDL_LIB=weaklysuperviseddl_amd/csrc/exp/libwsdl_<name>.so
set -e
name="$1"; shift
cd "$(dirname "$0")/.."
C=weaklysuperviseddl_amd/csrc
python -m weaklysuperviseddl_amd._build >/dev/null
mkdir -p $C/exp
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $C/conv_igemm.hip -o $C/exp/conv_igemm_$name.o
objs=""
for f in common plan norm_pool resample_loss layercam_optim lovasz components; do objs="$objs $C/build/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $C/exp/libwsdl_$name.so $C/exp/conv_igemm_$name.o $objs
rm -f $C/exp/conv_igemm_$name.o
echo "built $C/exp/libwsdl_$name.so"
