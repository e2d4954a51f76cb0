#!/bin/bash
# round 6 final collection: profiles (scripts/collect_profiles.sh) + the whole GPU suite
cd /root/repo
bash scripts/collect_profiles.sh > gpurun_out/collect.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/profiles_raw/pytest_gpu_all.txt 2>&1
tail -3 gpurun_out/profiles_raw/pytest_gpu_all.txt
ls gpurun_out/profiles_raw | head -50
