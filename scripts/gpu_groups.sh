#!/bin/bash
cd /root/repo
for g in 2 3 4 2 4; do
  LHW_ROLLOUT_GROUPS=$g timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('groups=$g value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4))"
done
