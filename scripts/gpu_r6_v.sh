#!/bin/bash
# round 6: a hidden layer's HBM stores issued one per K step of the next layer's products (HoldTile) vs the burst in the epilogue (head)
cd /root/repo; mkdir -p gpurun_out/r6v
V=/root/repo/learninghumanoidwalking_amd/variants
timeout 900 python -m pytest tests/test_mlp_strip_gpu.py tests/test_ppo_gpu.py tests/test_iteration_gpu.py -m gpu -x -q 2>&1 | tail -2
for L in "" head; do
  echo "variant ${L:-intree}"; ( [ -n "$L" ] && export LHW_LIB=$V/liblhw_$L.so; timeout 100 python scripts/strip_bench.py 32768; timeout 100 python scripts/strip_bench.py 65536 ) 2>/dev/null | grep " strip"
done | tee gpurun_out/r6v/strip.txt
for A in "32768 fwd" "32768 bwd"; do
  LHW_LIB=$V/liblhw_clock.so timeout 120 python scripts/strip_clock.py $A 2>&1 | grep -v "^Using\|amdgpu.ids"
done | tee gpurun_out/r6v/strip_clock.txt
mv $V/liblhw_clock.so /tmp/
bash scripts/gpu_ab.sh r6v/ab --steps 8 --warmup 3 | tee gpurun_out/r6v/ab.txt
