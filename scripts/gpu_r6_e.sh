#!/bin/bash
# round 6: full GPU suite on the current tree, then the profile collection (bench lines of every env, kernel stats, counter passes, training log)
cd /root/repo; mkdir -p gpurun_out/r6e
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6e/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6e/pytest_gpu.txt
bash scripts/collect_profiles.sh > gpurun_out/r6e/collect.log 2>&1
tail -3 gpurun_out/r6e/collect.log
