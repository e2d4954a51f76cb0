#!/bin/bash
# same-box A/B: --fp16 update with / without the fused half-input skinny weight gradients
cd /root/repo; mkdir -p gpurun_out/r6h
for V in 1 0 1 0 1 0; do
  LHW_WGRAD_FUSED=$V timeout 300 python bench.py --env h1 --num-envs 8192 --steps 3 --warmup 2 --no-cpu-baseline --fp16 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("h1 8192 --fp16 fused_skinny=$V", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4))
PY
done | tee gpurun_out/r6h/runs.txt
