#!/bin/bash
cd /root/repo
timeout 400 python -m pytest tests/test_h1_gpu.py tests/test_h1_walk_gpu.py tests/test_jvrc_gpu.py tests/test_fullsize_gpu.py tests/test_jvrc_step_gpu.py -x -q 2>&1 | tail -3
for e in "h1 8192" "h1_walk 8192" "jvrc_walk 4096" "jvrc_step 4096"; do set -- $e
  timeout 200 python bench.py --env $1 --num-envs $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4), d['stepper_counters']['one_env_per_wave_reruns'])"
done
