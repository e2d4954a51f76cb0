#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/chk
for i in 1 2 3; do timeout 100 python bench.py --env cartpole --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/chk/cartpole_$i.json; python -c "
import json; d=json.load(open('gpurun_out/chk/cartpole_$i.json')); print('cartpole', round(d['value']), d['sample_s_per_iter'], d['optimize_s_per_iter'])"; done
for i in 1 2; do timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --task-hook walking 2>/dev/null | tail -1 > gpurun_out/chk/plugin_$i.json; python -c "
import json; d=json.load(open('gpurun_out/chk/plugin_$i.json')); print('plugin', round(d['value']), d['sample_s_per_iter'], d['optimize_s_per_iter'])"; done
