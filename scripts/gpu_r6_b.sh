#!/bin/bash
# round 6, second GPU call: probe (exact inputs), full GPU suite, A/B (pointer tables vs member tables; dense-solve predicates), phase profile, PMC passes
cd /root/repo; mkdir -p gpurun_out/r6b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/dpp_probe.hip -o /tmp/dpp_probe 2>/dev/null && timeout 60 /tmp/dpp_probe > gpurun_out/r6b/dpp_probe.txt 2>&1
tail -3 gpurun_out/r6b/dpp_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6b/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6b/pytest_gpu.txt
bash scripts/gpu_ab.sh r6b/ab --steps 6 --warmup 3 2>&1 | tail -12
timeout 300 python scripts/jvrc_phase_profile.py 4096 jvrc_walk > gpurun_out/r6b/phase_cycles.txt 2>&1
cat gpurun_out/r6b/phase_cycles.txt
bash scripts/gpu_pmc.sh traffic jvrc_walk > gpurun_out/r6b/pmc.log 2>&1
cp gpurun_out/pmc/jvrc_walk_traffic.csv gpurun_out/r6b/
tail -40 gpurun_out/r6b/pmc.log
