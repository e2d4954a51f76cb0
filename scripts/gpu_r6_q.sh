#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6q
for S in 1 2 3; do timeout 200 python scripts/jvrc_phase_profile.py 4096 jvrc_step $S 2>/dev/null | grep -v "^Using"; done | tee gpurun_out/r6q/jvrc_step_phase.txt
timeout 200 python scripts/jvrc_phase_profile.py 4096 jvrc_walk 1 2>/dev/null | grep -v "^Using" | tee gpurun_out/r6q/jvrc_walk_phase.txt
