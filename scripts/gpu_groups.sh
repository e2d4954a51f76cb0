#!/bin/bash
# rollout groups (x hardware queues with "q": GPU_MAX_HW_QUEUES; HIP streams beyond it share a hardware queue and serialise)
cd /root/repo
for rep in 1 2; do for g in ${GROUPS_LIST:-1 2}; do
  LHW_ROLLOUT_GROUPS=$g timeout 200 python bench.py --env ${ENV:-jvrc_walk} --num-envs ${NENV:-4096} --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${ENV:-jvrc_walk} groups=$g value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4))"
done; done
