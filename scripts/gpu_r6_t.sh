#!/bin/bash
# round 6: 32-row slabs, four workgroups per CU with staggered issue priorities (LHW_STRIP_UPDATE_SHAPE=mid) vs 64-row slabs (big)
cd /root/repo; mkdir -p gpurun_out/r6t
for SH in big mid mid-flat; do
  echo "shape $SH"; ( LHW_STRIP_UPDATE_SHAPE=$SH timeout 100 python scripts/strip_bench.py 65536; LHW_STRIP_UPDATE_SHAPE=$SH timeout 100 python scripts/strip_bench.py 32768 ) 2>/dev/null | grep " strip"
done | tee gpurun_out/r6t/strip.txt
LHW_STRIP_UPDATE_SHAPE=mid timeout 900 python -m pytest tests/test_mlp_strip_gpu.py tests/test_ppo_gpu.py tests/test_iteration_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for SH in big mid mid-flat; do
  LHW_STRIP_UPDATE_SHAPE=$SH timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk shape=$SH", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4))
PY
done; done | tee gpurun_out/r6t/runs.txt
