#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4f; mkdir -p $O
for G in 1 2; do
  GPU_MAX_HW_QUEUES=8 LHW_ROLLOUT_GROUPS=$G timeout 300 python bench.py --env jvrc_step --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_step_g$G.json
  python - <<PY
import json
d=json.loads(open("$O/bench_step_g$G.json").read())
print("groups $G", round(d["value"]), "sample", round(d["sample_s_per_iter"],3))
PY
done
