#!/bin/bash
# kernel-trace stats of the default bench command (the control-step kernel's average must agree with the bench line's HIP events)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r4kstats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python /root/repo/bench.py --no-cpu-baseline > $O/kt.log 2>&1
cp /tmp/kt/*/*kernel_stats.csv $O/jvrc_walk_kernel_stats.csv
grep '^{' $O/kt.log | tail -1 > $O/bench_jvrc_walk_under_rocprof.json
grep "humanoid_kernel<0, 1, 32>" $O/jvrc_walk_kernel_stats.csv | cut -c1-60,200-330
python - <<PY
import json
d=json.loads(open("$O/bench_jvrc_walk_under_rocprof.json").read()); r=d["roofline"]
print(round(d["value"]), d["sample_s_per_iter"], r["overlapped"]["avg_launch_ms"], r["overlapped"]["launches"], r["avg_launch_ms"])
PY
