#!/bin/bash
# round 6, third GPU call: Jacobian row in registers A/B; the other envs against the round-5 library
cd /root/repo; mkdir -p gpurun_out/r6c
bash scripts/gpu_ab.sh r6c/ab --steps 6 --warmup 3 2>&1 | tail -8
for E in "h1 8192" "h1_walk 8192" "jvrc_step 4096"; do
  set -- $E
  for L in intree r05 intree r05; do
    if [ $L = intree ]; then unset LHW_LIB; else export LHW_LIB=/root/repo/learninghumanoidwalking_amd/variants/liblhw_$L.so; fi
    timeout 300 python bench.py --env $1 --num-envs $2 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("$1 $2 $L", round(d['value']), round(d['sample_s_per_iter'],4), round(d['optimize_s_per_iter'],4), d['stepper_counters'].get('contact_overflow_steps'))
PY
  done
done | tee gpurun_out/r6c/other_envs.txt
