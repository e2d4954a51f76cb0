#!/bin/bash
# the GPU suite under the poison build: LDS, device allocations and the output buffers 0xFF / NaN-filled (a read of unwritten memory shows as a NaN)
cd /root/repo; mkdir -p gpurun_out/poison
LHW_POISON=1 LHW_LIB=/root/repo/learninghumanoidwalking_amd/variants/liblhw_poison.so timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/poison/pytest_gpu_poison.txt 2>&1
tail -4 gpurun_out/poison/pytest_gpu_poison.txt
