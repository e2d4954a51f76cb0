#!/bin/bash
# rollout groups x hardware queues (GPU_MAX_HW_QUEUES: HIP streams beyond it share a hardware queue and serialise)
cd /root/repo
for q in 4 8; do for g in 2 3 4; do
  GPU_MAX_HW_QUEUES=$q LHW_ROLLOUT_GROUPS=$g timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues=$q groups=$g value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4))"
done; done
