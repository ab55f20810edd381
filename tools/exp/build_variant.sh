#!/bin/bash
# experiment (LAB) build of the library — environment knobs, timing ablations, the LDS-staged Eq. 8 variants: tools/exp/build_variant.sh <name> [-Dflags...]  ->  tools/exp/lib_<name>.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -DDIGAT_LAB "$@" -o tools/exp/lib_$NAME.so digat_amd/csrc/digat_kernels.hip
echo tools/exp/lib_$NAME.so
