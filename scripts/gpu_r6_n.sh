#!/bin/bash
# round 6: why is the wide weight gradient slower inside the update than alone?  kernel stats of the update with and without it; strip clock
cd /root/repo; mkdir -p gpurun_out/r6n
V=/root/repo/learninghumanoidwalking_amd/variants
for A in "32768 fwd" "32768 bwd"; do
  LHW_LIB=$V/liblhw_clock.so timeout 120 python scripts/strip_clock.py $A 2>&1 | grep -v "^Using\|amdgpu.ids"
done | tee gpurun_out/r6n/strip_clock.txt
cd /tmp && export TMPDIR=/tmp
for W in 1 0; do
  rm -rf /tmp/kt; LHW_WGRAD_WIDE=$W timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
  cp /tmp/kt/*/*kernel_stats.csv /root/repo/gpurun_out/r6n/kernel_stats_wide$W.csv
  grep '^{' /tmp/kt.log | tail -1 > /root/repo/gpurun_out/r6n/bench_wide$W.json
done
cd /root/repo
for rep in 1 2 3; do
for W in 1 0; do
  LHW_WGRAD_WIDE=$W timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk wide=$W", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4))
PY
done; done | tee gpurun_out/r6n/runs.txt
