#!/bin/bash
# round 6: start skew of the second-slot workgroups of the strip kernels: sweep alone, timeline of one setting, then inside the update
cd /root/repo; mkdir -p gpurun_out/r6p
for SK in 0 6 10 14 18 24; do
  echo "skew $SK us"; ( LHW_STRIP_SKEW=$SK timeout 100 python scripts/strip_bench.py 65536; LHW_STRIP_SKEW=$SK timeout 100 python scripts/strip_bench.py 32768 ) 2>/dev/null | grep " strip"
done | tee gpurun_out/r6p/sweep.txt
V=/root/repo/learninghumanoidwalking_amd/variants
for A in "32768 fwd" "65536 fwd"; do
  LHW_STRIP_SKEW=14 LHW_LIB=$V/liblhw_clock.so timeout 120 python scripts/strip_clock.py $A 2>&1 | grep -v "^Using\|amdgpu.ids"
done | tee gpurun_out/r6p/strip_clock_skew14.txt
mv $V/liblhw_clock.so /tmp/
for rep in 1 2; do
for SK in 0 8 14 20; do
  LHW_STRIP_SKEW=$SK timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("jvrc_walk skew=$SK", round(d['value']), "sample", round(d['sample_s_per_iter'],4), "opt", round(d['optimize_s_per_iter'],4))
PY
done; done | tee gpurun_out/r6p/runs.txt
