#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_jvrc_step_gpu.py -m gpu -q 2>&1 | tail -3 > $O/pytest_step.txt
timeout 600 python bench.py --env jvrc_step --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_step.json 2> $O/bench_step.err
timeout 300 python scripts/step_tail_by_mode.py 4096 60 > $O/step_tail.txt 2>&1
tail -2 $O/pytest_step.txt; python - <<PY
import json
d=json.loads(open("$O/bench_step.json").read().strip().splitlines()[-1])
print("bench_step", round(d["value"]), "sample", round(d["sample_s_per_iter"],3), d["stepper_counters"]["contact_overflow_steps"], d["stepper_counters"]["diverged_env_steps"])
PY
head -6 $O/step_tail.txt
