#!/bin/bash
# time stepper variants (learninghumanoidwalking_amd/variants/liblhw_*.so) against the in-tree library, same box, interleaved
OUT=/root/repo/gpurun_out/var
mkdir -p $OUT
cd /root/repo
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'value %.0f sample %.3f opt %.3f launch_ms %.2f iso %.3f' % (d['value'], d['sample_s_per_iter'], d['optimize_s_per_iter'], r['avg_launch_ms'], r['isolated']['launch_ms']))"; }
for rep in 1 2; do
  unset LHW_LIB; run default
  LHW_ROLLOUT_PERSISTENT=0 run default_launch_per_step
  for f in learninghumanoidwalking_amd/variants/liblhw_*.so; do
    export LHW_LIB=/root/repo/$f; run $(basename $f .so)
  done
done
