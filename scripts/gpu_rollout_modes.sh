#!/bin/bash
# resident rollout (lhw_env_rollout) vs the launch-per-step pipeline -- parity tests, then same-box interleaved A/B benches.
# $1 = output tag
TAG=${1:-r5_resident}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_rollout_resident_gpu.py tests/test_iteration_gpu.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
for i in 1 2; do
  for MODE in steps resident; do
    LHW_ROLLOUT_MODE=$MODE timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>$OUT/bench_${MODE}_$i.err | tail -1 > $OUT/bench_jvrc_walk_${MODE}_$i.json
  done
done
for E in h1 h1_walk jvrc_step; do
  NE=8192; [ $E = jvrc_step ] && NE=4096
  for MODE in steps resident; do
    LHW_ROLLOUT_MODE=$MODE timeout 300 python bench.py --env $E --num-envs $NE --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${E}_${MODE}.json
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = d["roofline"]
    print(os.path.basename(f), "value", round(d["value"]), "sample", round(d["sample_s_per_iter"], 4), "opt", round(d["optimize_s_per_iter"], 4), "mode", r.get("rollout_mode"),
          "iso_ms", round(r["avg_launch_ms"], 4), {k: v for k, v in d["stepper_counters"].items() if k != "note"})
PY
