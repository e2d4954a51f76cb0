#!/bin/bash
# round 6: GPU suite on the tree with cylinder geoms, then the headline and jvrc_step lines
cd /root/repo; mkdir -p gpurun_out/r6k
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6k/pytest.txt 2>&1
tail -6 gpurun_out/r6k/pytest.txt
for E in jvrc_walk jvrc_step; do
  timeout 300 python bench.py --env $E --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6k/bench_$E.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r6k/bench_$E.json'))
print("$E", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), "frac", round(d['roofline']['frac'],4))
PY
done | tee gpurun_out/r6k/runs.txt
