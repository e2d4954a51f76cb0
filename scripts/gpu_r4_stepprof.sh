#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4e; mkdir -p $O
for seed in 12 7 3; do timeout 200 python scripts/jvrc_phase_profile.py 4096 jvrc_step $seed > $O/phase_step_seed$seed.txt 2>&1; done
timeout 200 python scripts/jvrc_phase_profile.py 4096 jvrc_walk 1 > $O/phase_walk.txt 2>&1
cat $O/phase_step_seed12.txt $O/phase_step_seed7.txt $O/phase_step_seed3.txt $O/phase_walk.txt
