#!/bin/bash
# round 6, last run on the final tree: the whole GPU suite, smoke(), the headline bench line (reads the committed counter passes)
cd /root/repo; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu_all.txt 2>&1; tail -3 gpurun_out/final/pytest_gpu_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/bench_jvrc_walk_1gpu.json
python -c "
import json; d=json.load(open('gpurun_out/final/bench_jvrc_walk_1gpu.json')); r=d['roofline']; print(round(d['value']), d['sample_s_per_iter'], d['optimize_s_per_iter'], r['frac'], r['executed']['valu_instructions_per_env_substep'], d['cpu_baseline']['value'])"
