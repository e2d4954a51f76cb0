#!/bin/bash
# sub-phase clock of one stepper phase: gpu_r4_fine.sh "PHASES" ENV "SEEDS"   (variants liblhw_fine<phase>.so built beforehand)
cd "$(dirname "$0")/.."
O=gpurun_out/r4fine; mkdir -p $O
for ph in ${1:-3}; do for seed in ${3:-1}; do LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/liblhw_fine$ph.so timeout 200 python scripts/fine_phase_profile.py 4096 ${2:-jvrc_walk} $seed 2>&1 | grep -v amdgpu.ids; done; done > $O/fine_${2:-jvrc_walk}.txt 2>&1
cat $O/fine_${2:-jvrc_walk}.txt
