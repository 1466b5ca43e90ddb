#!/bin/bash
# Kernel-experiment builds: compiles ONE source of the library with extra -D flags and links it with the production objects
# of the others into build_tools/libacrmi_<tag>.so (git-ignored; travels with gpurun).  Use: ACRMI_LIB=build_tools/libacrmi_<tag>.so
#   tools/exp_build.sh <tag> <source.hip> [-DFLAG ...]
set -e
TAG=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/arbitrary-hands-3d-reconstruction_amd/csrc
mkdir -p $R/build_tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $C/$SRC -o $R/build_tools/${SRC%.hip}_$TAG.o
OBJS=""
for o in $C/*.o; do
  if [ "$(basename $o)" != "${SRC%.hip}.o" ]; then OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_tools/libacrmi_$TAG.so $OBJS $R/build_tools/${SRC%.hip}_$TAG.o
echo $R/build_tools/libacrmi_$TAG.so
