#!/bin/bash
# rollout-group sweep (LHW_ROLLOUT_GROUPS) on the final kernels: env, envs, groups -> env-steps/s, sample s, update s
cd "$(dirname "$0")/.."
O=gpurun_out/r4groups; mkdir -p $O
run() { LHW_ROLLOUT_GROUPS=$3 timeout 200 python bench.py --env $1 --num-envs $2 --steps ${4:-5} --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, 'groups', $3, round(d['value']), round(d['sample_s_per_iter'],4), round(d['optimize_s_per_iter'],4))"; }
{
for g in 1 2 3 4; do run jvrc_walk 4096 $g; done
for g in 1 2; do run jvrc_walk 2048 $g; done
for g in 1 2; do run jvrc_walk 1024 $g; done
for g in 2 4; do run jvrc_walk 8192 $g 3; done
for g in 2 3 4; do run h1 8192 $g 3; done
for g in 1 2; do run h1 4096 $g 3; done
for g in 2 4; do run jvrc_step 4096 $g 3; done
} > $O/groups.txt 2>&1
cat $O/groups.txt
