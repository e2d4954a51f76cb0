#!/bin/bash
cd /root/repo
for Q in 8 16; do for G in 4 8; do
  GPU_MAX_HW_QUEUES=$Q LHW_ROLLOUT_GROUPS=$G python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('Q=$Q G=$G value %.0f sample_s %.3f launch_ms %.3f wall_ms/step %.3f'%(d['value'], d['sample_s_per_iter'], d['roofline']['avg_launch_ms'], d['roofline']['aggregate']['wall_ms_per_control_step']))"
done; done
