#!/bin/bash
# round 6: (1) phase timeline of the strip kernels (clock build), (2) jvrc_step before / after the cylinder commit on one box
cd /root/repo; mkdir -p gpurun_out/r6l
V=/root/repo/learninghumanoidwalking_amd/variants
for A in "32768 fwd" "65536 fwd" "32768 bwd" "65536 bwd"; do
  LHW_LIB=$V/liblhw_clock.so timeout 120 python scripts/strip_clock.py $A 2>&1 | grep -v "^Using"
done | tee gpurun_out/r6l/strip_clock.txt
mv $V/liblhw_clock.so /tmp/
bash scripts/gpu_ab.sh r6l/ab --env jvrc_step --steps 3 --warmup 1 | tee gpurun_out/r6l/ab.txt
timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -3
