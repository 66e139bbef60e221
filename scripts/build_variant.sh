#!/bin/bash
# scripts/build_variant.sh <name> [-Dmacro=value ...] -- A/B build of the HIP library into polaris_amd/lib/exp/<name>.so
# (picked up with POLARIS_HIP_LIB=polaris_amd/lib/exp/<name>.so; lib/ is git-ignored but travels with gpurun).  Only the tracer's
# translation unit is rebuilt with the macros; the device BVH builder's object is the in-tree one (run `make -C polaris_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $ROOT/polaris_amd/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function \
  -I$ROOT/include -I$ROOT/polaris_amd/csrc "$@" -c $ROOT/polaris_amd/csrc/polaris_hip.hip -o $ROOT/polaris_amd/lib/exp/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $ROOT/polaris_amd/lib/exp/$name.o $ROOT/polaris_amd/lib/obj/bvh_build.o -o $ROOT/polaris_amd/lib/exp/$name.so
rm -f $ROOT/polaris_amd/lib/exp/$name.o
