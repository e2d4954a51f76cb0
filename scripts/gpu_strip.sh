#!/bin/bash
# strip kernels: parity tests, then the update's time with per-layer GEMMs (0), strip kernels in the update (1), and in the
# rollout inference as well (2)
cd /root/repo
timeout 300 python -m pytest tests/test_mlp_strip_gpu.py tests/test_ppo_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -5
for rep in 1 2; do
for m in 0 1 2; do
  LHW_MLP_STRIP=$m timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('strip=$m value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4))"
done; done
