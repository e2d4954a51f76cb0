#!/bin/bash
# round-2 GPU pass A: parity of the group-width-generic stepper on hardware + first timings
set -u
OUT=/root/repo/gpurun_out/r2b
mkdir -p $OUT
cd /root/repo
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/bench_fast.err | tail -1 > $OUT/bench_walk_fast.json
LHW_ONE_ENV_PER_WAVE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_walk_w64.json
python bench.py --env h1 --num-envs 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_h1_fast.json
python bench.py --env jvrc_step --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_step.json
python scripts/jvrc_phase_profile.py 4096 > $OUT/phase_fast.txt 2>&1
LHW_ONE_ENV_PER_WAVE=1 python scripts/jvrc_phase_profile.py 4096 > $OUT/phase_w64.txt 2>&1
python - <<'PY' > $OUT/summary.txt
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r2b/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f'%d['value'], 'sample_s %.3f opt_s %.3f'%(d['sample_s_per_iter'], d['optimize_s_per_iter']), 'launch_ms %.3f wall_ms/step %.3f'%(r['avg_launch_ms'], r['aggregate']['wall_ms_per_control_step']))
    except Exception as e: print(f, 'ERR', e)
PY
cat $OUT/summary.txt; cat $OUT/phase_fast.txt | head -20
