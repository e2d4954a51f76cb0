#!/bin/bash
# The resident rollout's job queue (LHW_ROLLOUT_CHUNK, csrc/lhw_humanoid_rollout.hip) on / off, same box, interleaved.
# usage: gpu_queue_ab.sh TAG
TAG=${1:-queue_ab}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
run() {  # name chunk bench-args...
  local n=$1 c=$2; shift 2
  LHW_ROLLOUT_CHUNK=$c timeout 300 python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${n}_chunk${c}_$REP.json
}
for REP in 1 2; do
  for c in ${CHUNKS:-0 10}; do run jvrc_step $c --env jvrc_step; done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), "value", round(d["value"]), "sample", round(d["sample_s_per_iter"], 4), "opt", round(d["optimize_s_per_iter"], 4), d["roofline"].get("rollout_mode"))
PY
