#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6x
V=/root/repo/learninghumanoidwalking_amd/variants
for P in 4 2 7 3; do LHW_LIB=$V/liblhw_fine$P.so timeout 200 python scripts/fine_phase_profile.py 4096 jvrc_walk 1 2>/dev/null | grep -v "^Using"; done | tee gpurun_out/r6x/fine.txt
