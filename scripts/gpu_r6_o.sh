#!/bin/bash
# round 6: wide weight gradient, register-buffer depth 16 vs 32 rows: alone (warm / cold operands) and inside the update
cd /root/repo; mkdir -p gpurun_out/r6o
for KS in 16 32; do for R in 32768 65536; do
  LHW_WGRAD_WIDE_KS=$KS timeout 200 python scripts/gemm_bench.py $R 2>/dev/null | grep -E "^dW2 (LDS|wide)|^rows"
done; done | tee gpurun_out/r6o/gemm.txt
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k wide 2>&1 | tail -2
for rep in 1 2 3; do
for V in "1 32" "0 32" "1 16"; do
  set -- $V
  LHW_WGRAD_WIDE=$1 LHW_WGRAD_WIDE_KS=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk wide=$1 ks=$2", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4))
PY
done; done | tee gpurun_out/r6o/runs.txt
