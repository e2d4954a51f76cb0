#!/bin/bash
# round 6: the stepping task's rollout kernels as their own translation unit under the max-ILP scheduling strategy (in-tree) vs one TU (head)
cd /root/repo; mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
bash scripts/gpu_ab.sh r6y/ab_step --env jvrc_step --steps 3 --warmup 1 | tee gpurun_out/r6y/ab_step.txt
bash scripts/gpu_ab.sh r6y/ab --steps 6 --warmup 3 | tee gpurun_out/r6y/ab.txt
