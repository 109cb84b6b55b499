#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags ...]   -> tools/bin/variants/libmsplat_<name>.so (same ABI; load it with MSPLAT_LIB_PATH)
cd "$(dirname "$0")/.."
mkdir -p tools/bin/variants
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function "$@" \
  -o tools/bin/variants/libmsplat_$N.so splatapult_amd/csrc/msplat_device.hip splatapult_amd/csrc/msplat_group.hip \
  splatapult_amd/host/gaussian_scene.cpp splatapult_amd/host/scene_config.cpp splatapult_amd/host/point_scene.cpp -lpthread
