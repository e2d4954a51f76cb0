#!/bin/bash
# round 6, first GPU call: hardware probe of v_fmac_f64_dpp, the stepper parity tests on the new kernels, same-box A/B against the round-5 library
cd /root/repo; mkdir -p gpurun_out/r6a
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/dpp_probe.hip -o /tmp/dpp_probe 2>/dev/null && timeout 60 /tmp/dpp_probe > gpurun_out/r6a/dpp_probe.txt 2>&1
cat gpurun_out/r6a/dpp_probe.txt | tail -5
timeout 1500 python -m pytest tests/test_jvrc_gpu.py tests/test_rollout_resident_gpu.py tests/test_wide_batch_gpu.py tests/test_h1_gpu.py tests/test_jvrc_step_gpu.py tests/test_freerun_gpu.py -m gpu -x -q > gpurun_out/r6a/pytest_stepper.txt 2>&1
tail -5 gpurun_out/r6a/pytest_stepper.txt
bash scripts/gpu_ab.sh r6a/ab --steps 6 --warmup 3 2>&1 | tail -12
