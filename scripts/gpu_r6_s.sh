#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6s
timeout 900 python -m pytest tests/test_jvrc_step_gpu.py -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_ab.sh r6s/ab --env jvrc_step --steps 3 --warmup 1 | tee gpurun_out/r6s/ab.txt
