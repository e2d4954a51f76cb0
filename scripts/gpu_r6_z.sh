#!/bin/bash
# round 6: the strips' weight transposes and the loss statistics' reduction moved to the side stream (in-tree) vs on the actor's chain (head)
cd /root/repo; mkdir -p gpurun_out/r6z
timeout 1200 python -m pytest tests/test_ppo_gpu.py tests/test_iteration_gpu.py tests/test_distributed_gpu.py tests/test_task_hook_gpu.py -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_ab.sh r6z/ab --steps 10 --warmup 3 | tee gpurun_out/r6z/ab.txt
