#!/bin/bash
# round 4: the stepping task on the reference terrain (many-contact path): parity tests + throughput
cd "$(dirname "$0")/.."
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests/test_jvrc_step_gpu.py tests/test_entry_gpu.py -m gpu -q 2>&1 | tail -15 > $O/pytest_step.txt
timeout 600 python bench.py --env jvrc_step --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_step.json 2> $O/bench_step.err
LHW_STEP_NO_BIG=1 timeout 600 python bench.py --env jvrc_step --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_step_nobig.json 2> $O/bench_step_nobig.err
tail -5 $O/pytest_step.txt
for f in bench_step bench_step_nobig; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "sample", round(d["sample_s_per_iter"],3), d["stepper_counters"]["contact_overflow_steps"], d["stepper_counters"]["diverged_env_steps"])
except Exception as e: print("$f failed", e)
PY
done
