#!/bin/bash
# builds hope_amd/libhope_env_<tag>.so from the sources of git revision REV (A/B timing of two kernel versions in one gpurun call:
# HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_<tag>.so):   bash tools/build_variant.sh REV TAG [extra hipcc flags]
set -e
REV=$1; TAG=$2; shift 2
D=$(mktemp -d)
mkdir -p $D/csrc $D/include
for f in $(git ls-tree --name-only $REV hope_amd/csrc/); do git show $REV:$f > $D/csrc/$(basename $f); done
git show $REV:include/hope_env.h > $D/include/hope_env.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wno-unused-value -pthread "$@" \
  -I$D/include -I$D/csrc $D/csrc/hope_env.hip $D/csrc/hope_rs.hip $D/csrc/hope_bev.hip $D/csrc/hope_scenegen.cpp -o hope_amd/libhope_env_$TAG.so
rm -rf $D; ls -la hope_amd/libhope_env_$TAG.so
