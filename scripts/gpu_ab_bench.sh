#!/bin/bash
# same-box A/B of the whole bench: in-tree library vs every learninghumanoidwalking_amd/variants/liblhw_*.so, interleaved
cd /root/repo
for rep in 1 2; do
  for f in "" learninghumanoidwalking_amd/variants/liblhw_*.so; do
    if [ -n "$f" ]; then export LHW_LIB=/root/repo/$f; else unset LHW_LIB; fi
    timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${f:-default}', 'value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4), 'iso', round(d['roofline']['avg_launch_ms'],4), 'ovl', round(d['roofline']['overlapped']['avg_launch_ms'],4))"
  done
done
