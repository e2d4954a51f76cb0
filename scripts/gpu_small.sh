#!/bin/bash
cd /root/repo
timeout 300 python -m pytest tests/test_mlp_strip_gpu.py tests/test_ppo_gpu.py tests/test_iteration_gpu.py tests/test_entry_gpu.py -x -q 2>&1 | tail -2
timeout 100 python scripts/chain_latency.py 2>&1 | grep -E "rows|inference chain" | tail -4
for i in 1 2; do timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4))"; done
