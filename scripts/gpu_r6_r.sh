#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6r
V=/root/repo/learninghumanoidwalking_amd/variants
for S in 1 2; do LHW_LIB=$V/liblhw_fine3.so timeout 200 python scripts/fine_phase_profile.py 4096 jvrc_step $S 2>/dev/null | grep -v "^Using"; done | tee gpurun_out/r6r/fine3.txt
LHW_LIB=$V/liblhw_fine3.so timeout 200 python scripts/fine_phase_profile.py 4096 jvrc_walk 1 2>/dev/null | grep -v "^Using" | tee -a gpurun_out/r6r/fine3.txt
