#!/bin/bash
# round-2 GPU pass D: persistent rollout kernel (tests first, then A/B against the launch-per-step rollout in one job)
set -u
OUT=/root/repo/gpurun_out/r2g
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_rollout_gpu.py -x -q -s > $OUT/pytest_rollout.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_rollout.log
tail -8 $OUT/pytest_rollout.log
for P in 1 0; do
  LHW_ROLLOUT_PERSISTENT=$P timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/bench_p$P.err | tail -1 > $OUT/bench_walk_p$P.json
done
LHW_ROLLOUT_PERSISTENT=1 timeout 600 python bench.py --env h1 --num-envs 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_h1_p1.json
python - <<'PY' > $OUT/summary.txt
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r2g/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f'%d['value'], 'sample_s %.3f opt_s %.3f'%(d['sample_s_per_iter'], d['optimize_s_per_iter']), 'launch_ms %.3f wall_ms/step %.3f'%(r['avg_launch_ms'], r['aggregate']['wall_ms_per_control_step']), 'iso', r['isolated'] and r['isolated']['launch_ms'], d['stepper_counters'])
    except Exception as e: print(f, 'ERR', e)
PY
cat $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
