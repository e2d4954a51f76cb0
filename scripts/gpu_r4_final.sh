#!/bin/bash
# round 4: final validation -- smoke, full GPU suite (product build; then LDS / allocations poisoned), two-rank test repeated, bench
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r4g}; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_full.txt
LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/liblhw_poison.so LHW_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_poison.txt
for i in $(seq 1 10); do timeout 300 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k two_ranks 2>&1 | tail -1; done > $O/dp_repeat.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err   # (the driver's command line)
tail -3 $O/smoke.txt; tail -2 $O/pytest_full.txt; tail -2 $O/pytest_poison.txt; sort $O/dp_repeat.txt | cut -c1-20 | uniq -c
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", round(d["value"]), "sample", round(d["sample_s_per_iter"],4), "opt", round(d["optimize_s_per_iter"],4), "iso_ms", round(r["avg_launch_ms"],4), "frac", round(r["frac"],4), "exec", r.get("executed_static",{}).get("frac_of_fp64_peak"), "upd", round(r["update"]["frac"],3), "ovl_ms", round(r["overlapped"]["avg_launch_ms"],4), d["stepper_counters"]["one_env_per_wave_reruns"])
PY
