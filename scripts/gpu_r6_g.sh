#!/bin/bash
# round 6: fp16 update with the fused half-input skinny weight gradients; box-box reciprocals (jvrc_step); full suite
cd /root/repo; mkdir -p gpurun_out/r6g
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6g/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6g/pytest_gpu.txt
for V in "h1 8192 --fp16" "h1 8192 --fp16" "jvrc_step 4096" "jvrc_step 4096"; do
  set -- $V
  timeout 300 python bench.py --env $1 --num-envs $2 --steps 3 --warmup 2 --no-cpu-baseline $3 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$1 $2 $3", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), "upd frac", round(d['roofline']['update']['frac'],4))
PY
done | tee gpurun_out/r6g/runs.txt
