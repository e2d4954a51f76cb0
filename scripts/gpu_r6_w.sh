#!/bin/bash
# round 6: same-box A/B of the in-tree stepper against learninghumanoidwalking_amd/variants/liblhw_head.so on three envs
cd /root/repo; mkdir -p gpurun_out/r6w
timeout 1200 python -m pytest tests/test_jvrc_gpu.py tests/test_h1_gpu.py tests/test_h1_walk_gpu.py tests/test_jvrc_step_gpu.py tests/test_wide_batch_gpu.py tests/test_rollout_resident_gpu.py -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_ab.sh r6w/ab --steps 8 --warmup 3 | tee gpurun_out/r6w/ab.txt
bash scripts/gpu_ab.sh r6w/ab_h1 --env h1 --num-envs 8192 --steps 3 --warmup 1 | tee gpurun_out/r6w/ab_h1.txt
bash scripts/gpu_ab.sh r6w/ab_step --env jvrc_step --steps 3 --warmup 1 | tee gpurun_out/r6w/ab_step.txt
