#!/bin/bash
# scripts/build_variant.sh NAME [-D...]: liblhw.so with extra compile flags -> learninghumanoidwalking_amd/variants/liblhw_NAME.so
# (kernel experiments: run with LHW_LIB=<that file>)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p learninghumanoidwalking_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -shared "$@" -o learninghumanoidwalking_amd/variants/liblhw_$NAME.so learninghumanoidwalking_amd/csrc/*.hip
echo built learninghumanoidwalking_amd/variants/liblhw_$NAME.so
