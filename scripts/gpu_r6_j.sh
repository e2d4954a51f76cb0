#!/bin/bash
# round 6: the optimiser step as one hipGraph launch -- its test, then the headline with and without it (same box, interleaved)
cd /root/repo; mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_iteration_gpu.py tests/test_ppo_gpu.py -m gpu -x -q > gpurun_out/r6j/pytest.txt 2>&1
tail -12 gpurun_out/r6j/pytest.txt
for V in 1 0 1 0 1 0; do
  LHW_PPO_GRAPH=$V timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk graph=$V", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), "upd frac", round(d['roofline']['update']['frac'],4))
PY
done | tee gpurun_out/r6j/runs.txt
