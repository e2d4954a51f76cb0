#!/bin/bash
# kernel-trace stats of the bench (one group and two groups) -> gpurun_out/trace/
OUT=/root/repo/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in 1 2; do
  rm -rf /tmp/kt$G
  LHW_ROLLOUT_GROUPS=$G timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$G -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_g$G.log 2>&1
  cp /tmp/kt$G/*/*kernel_stats.csv $OUT/kernel_stats_g$G.csv
  tail -1 $OUT/bench_g$G.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G=$G sample_s %.3f opt_s %.3f launch_ms %.3f isolated %s'%(d['sample_s_per_iter'], d['optimize_s_per_iter'], d['roofline']['avg_launch_ms'], d['roofline']['isolated']))"
  head -12 $OUT/kernel_stats_g$G.csv | cut -c1-150
done
