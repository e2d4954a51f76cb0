#!/bin/bash
# Same-box interleaved A/B of the in-tree library against every variant under learninghumanoidwalking_amd/variants/
# (scripts/build_variant.sh, or a copy of an earlier liblhw.so): bench.py rollout + update times.  usage: gpu_ab.sh TAG [bench args...]
TAG=${1:-ab}; shift
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
ARGS=${@:---steps 5 --warmup 2}
for rep in 1 2; do
  unset LHW_LIB; timeout 300 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_intree_$rep.json
  for f in learninghumanoidwalking_amd/variants/liblhw_*.so; do
    [ -f $f ] || continue
    n=$(basename $f .so)
    LHW_LIB=/root/repo/$f timeout 300 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${n}_$rep.json
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), "value", round(d["value"]), "sample", round(d["sample_s_per_iter"], 4), "opt", round(d["optimize_s_per_iter"], 4), d["roofline"].get("rollout_mode"),
          {k: v for k, v in d["stepper_counters"].items() if k != "note"})
PY
