#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4fine; mkdir -p $O
for ph in 3; do LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/liblhw_fine$ph.so timeout 200 python scripts/fine_phase_profile.py 4096 jvrc_walk; done > $O/fine.txt 2>&1
cat $O/fine.txt
