#!/bin/bash
# round 6: LLVM scheduling strategies per stepper translation unit (in-tree) vs the build before (head): the whole GPU suite, then every env
cd /root/repo; mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
bash scripts/gpu_ab.sh r6y/ab --steps 6 --warmup 3 | tee gpurun_out/r6y/ab.txt
bash scripts/gpu_ab.sh r6y/ab_h1 --env h1 --num-envs 8192 --steps 3 --warmup 1 | tee gpurun_out/r6y/ab_h1.txt
bash scripts/gpu_ab.sh r6y/ab_h1w --env h1_walk --num-envs 8192 --steps 3 --warmup 1 | tee gpurun_out/r6y/ab_h1w.txt
bash scripts/gpu_ab.sh r6y/ab_step --env jvrc_step --steps 3 --warmup 1 | tee gpurun_out/r6y/ab_step.txt
