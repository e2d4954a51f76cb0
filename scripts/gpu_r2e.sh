#!/bin/bash
# round-2 GPU pass E: bitwise reproducibility of the stepper between its two instantiations (-ffp-contract), A/B in one job
set -u
OUT=/root/repo/gpurun_out/r2h
mkdir -p $OUT
cd /root/repo
for V in default on; do
  if [ $V = default ]; then unset LHW_LIB; else export LHW_LIB=/root/repo/learninghumanoidwalking_amd/variants/liblhw_$V.so; fi
  timeout 900 python -m pytest tests/test_rollout_gpu.py -q -s > $OUT/pytest_rollout_$V.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_rollout_$V.log
  grep -E "AssertionError|passed|failed|bitwise|re-run" $OUT/pytest_rollout_$V.log | head -12
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/bench_$V.err | tail -1 > $OUT/bench_walk_$V.json
done
export LHW_LIB=/root/repo/learninghumanoidwalking_amd/variants/liblhw_on.so
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_on.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_on.log
tail -8 $OUT/pytest_on.log
python - <<'PY' > $OUT/summary.txt
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r2h/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f'%d['value'], 'sample_s %.3f opt_s %.3f'%(d['sample_s_per_iter'], d['optimize_s_per_iter']), 'launch_ms %.3f'%(r['avg_launch_ms']), 'iso', r['isolated'] and r['isolated']['launch_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
cat $OUT/summary.txt
