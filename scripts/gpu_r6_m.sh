#!/bin/bash
# round 6: wide weight gradient without LDS, ReLU masks as bits, batched slab staging -- tests, micro-benchmarks, same-box A/B of the update
cd /root/repo; mkdir -p gpurun_out/r6m
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_mlp_strip_gpu.py tests/test_ppo_gpu.py tests/test_iteration_gpu.py tests/test_model_variants_gpu.py -m gpu -x -q > gpurun_out/r6m/pytest.txt 2>&1
tail -8 gpurun_out/r6m/pytest.txt
( timeout 200 python scripts/gemm_bench.py 32768 2>/dev/null | grep -E "dW2|rows"; timeout 200 python scripts/gemm_bench.py 65536 2>/dev/null | grep -E "dW2|rows" ) | tee gpurun_out/r6m/gemm.txt
( timeout 100 python scripts/strip_bench.py 65536; timeout 100 python scripts/strip_bench.py 32768 ) 2>/dev/null | grep strip | tee gpurun_out/r6m/strip.txt
for rep in 1 2; do
for V in "1 1" "0 1" "1 0" "0 0"; do
  set -- $V
  LHW_WGRAD_WIDE=$1 LHW_STRIP_BITS=$2 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk wide=$1 bits=$2", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4), "upd frac", round(d['roofline']['update']['frac'],4))
PY
done; done | tee gpurun_out/r6m/runs.txt
timeout 300 python bench.py --env jvrc_step --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6m/bench_jvrc_step.json
python -c "
import json; d=json.load(open('gpurun_out/r6m/bench_jvrc_step.json')); print('jvrc_step', round(d['value']), d['sample_s_per_iter'])"
