#!/bin/bash
# development: build an A/B variant of libsigtk_gpu.so with extra -D flags; select it with SIGTK_AMD_LIB=<path>
# usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...]      (-DSGK_DEV=1: the instrumented build, csrc/event_args.h)
# Only event_kernels.hip and api.hip are recompiled; the other objects are the shipped build's (sigtk_amd/build/).
set -e
cd "$(dirname "$0")/.."
python -m sigtk_amd.build --variant "$@"
