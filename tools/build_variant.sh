#!/bin/bash
# development: build an A/B variant of libsigtk_gpu.so with extra -D flags; select it with SIGTK_AMD_LIB=<path>
# usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p sigtk_amd/_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared \
  -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-variable -Wno-c++20-extensions "$@" -o sigtk_amd/_variants/libsigtk_gpu_$tag.so \
  sigtk_amd/csrc/api.hip sigtk_amd/csrc/api_stat.hip sigtk_amd/csrc/event_kernels.hip sigtk_amd/csrc/stat_kernels.hip \
  sigtk_amd/csrc/misc_kernels.hip sigtk_amd/csrc/svb_kernels.hip sigtk_amd/csrc/ent_kernels.hip sigtk_amd/csrc/qts_kernels.hip \
  sigtk_amd/csrc/job.hip sigtk_amd/csrc/shims.hip
echo sigtk_amd/_variants/libsigtk_gpu_$tag.so
