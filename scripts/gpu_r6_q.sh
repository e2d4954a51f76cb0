#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6q
for S in 4 5 6 7 8 9 10 11; do timeout 200 python scripts/jvrc_phase_profile.py 4096 jvrc_step $S 2>/dev/null | grep -v "^Using"; done | tee gpurun_out/r6q/jvrc_step_phase_more.txt
