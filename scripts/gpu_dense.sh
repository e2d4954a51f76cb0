#!/bin/bash
cd /root/repo
timeout 300 python -m pytest tests/test_jvrc_gpu.py tests/test_h1_gpu.py tests/test_jvrc_step_gpu.py tests/test_freerun_gpu.py -x -q 2>&1 | tail -2
timeout 100 python scripts/tail_waves.py 150 2>&1 | grep -E "waves 2048|self-collision|episode end"
for i in 1 2; do timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4), 'iso', round(d['roofline']['avg_launch_ms'],4))"; done
timeout 120 python scripts/step_time.py 4096 2>/dev/null | tail -1
