#!/bin/bash
# round 5, second GPU pass: pipelined in-wave policy, task hook, -ffp-contract=on variant.  $1 = tag
TAG=${1:-r5_b}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_rollout_resident_gpu.py tests/test_task_hook_gpu.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
V=/root/repo/learninghumanoidwalking_amd/variants/liblhw_fpc_on.so
LHW_LIB=$V timeout 300 python scripts/resident_diff.py jvrc_step 97 12 3 2>&1 | grep -v "^Using\|amdgpu.ids" > $OUT/diff_step_fpc_on.txt; grep -c equal $OUT/diff_step_fpc_on.txt; grep differ $OUT/diff_step_fpc_on.txt | cut -c1-200
for i in 1 2; do
  LHW_ROLLOUT_MODE=resident timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_resident_$i.json
  LHW_LIB=$V LHW_ROLLOUT_MODE=resident timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_resident_fpcon_$i.json
done
LHW_ROLLOUT_MODE=steps timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_steps_1.json
for E in h1 h1_walk jvrc_step; do
  NE=8192; [ $E = jvrc_step ] && NE=4096
  LHW_ROLLOUT_MODE=resident timeout 300 python bench.py --env $E --num-envs $NE --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${E}_resident.json
done
LHW_LIB=$V LHW_ROLLOUT_MODE=resident timeout 300 python bench.py --env jvrc_step --num-envs 4096 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_jvrc_step_resident_fpcon.json
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
