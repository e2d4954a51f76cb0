#!/bin/bash
# round 6: JVRC init_noise / perturbation on the GPU, full suite, headline against the round-5 library on the same box
cd /root/repo; mkdir -p gpurun_out/r6i
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6i/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6i/pytest_gpu.txt
bash scripts/gpu_ab.sh r6i/ab --steps 6 --warmup 3 2>&1 | tail -6
