#!/bin/bash
# round 6: the fp16-storage GEMM -- its tests, then h1 @ 8192 --fp16 against float32 storage (LHW_FP16_STORAGE=0) and the float32 update
cd /root/repo; mkdir -p gpurun_out/r6f
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_ppo_gpu.py tests/test_rollout_resident_gpu.py::test_fp16_operand_policy_in_the_resident_rollout_matches_the_fp16_mfma_forward -m gpu -x -q > gpurun_out/r6f/pytest_fp16.txt 2>&1
tail -6 gpurun_out/r6f/pytest_fp16.txt
for V in "fp16 1" "fp16 0" "f32 1" "fp16 1" "fp16 0"; do
  set -- $V
  F=""; [ $1 = fp16 ] && F="--fp16"
  LHW_FP16_STORAGE=$2 timeout 300 python bench.py --env h1 --num-envs 8192 --steps 3 --warmup 2 --no-cpu-baseline $F 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("h1@8192 $1 storage_fp16=$2", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), "upd frac", round(d['roofline']['update']['frac'],3))
PY
done | tee gpurun_out/r6f/fp16_storage.txt
timeout 300 python scripts/gemm_bench.py 32768 > gpurun_out/r6f/gemm_shapes.txt 2>&1; tail -20 gpurun_out/r6f/gemm_shapes.txt
