#!/bin/bash
# round 4: full GPU suite + headline bench on the current tree
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_full.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest_full.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", round(d["value"]), "sample", round(d["sample_s_per_iter"],4), "opt", round(d["optimize_s_per_iter"],4), "iso_ms", r["avg_launch_ms"], "frac", round(r["frac"],4), "upd", r.get("update",{}).get("frac"), "ovl_ms", r["overlapped"]["avg_launch_ms"], d["stepper_counters"])
PY
