#!/bin/bash
# round 6: task-hook tests on the GPU and what the plug-in costs (fused / reward-only resident / own terminations)
cd /root/repo; mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests/test_task_hook_gpu.py tests/test_task_inputs_gpu.py tests/test_rollout_resident_gpu.py -m gpu -x -q > gpurun_out/r6d/pytest_hook.txt 2>&1
tail -15 gpurun_out/r6d/pytest_hook.txt
for H in none walking walking-own-done none walking; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --task-hook $H 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("task-hook $H", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), d['roofline'].get('rollout_mode'))
PY
done | tee gpurun_out/r6d/task_hook_cost.txt
