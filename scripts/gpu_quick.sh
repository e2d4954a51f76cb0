#!/bin/bash
# quick GPU check of a stepper change: jvrc GPU tests, short bench, phase profile, SQ counters.  $1 = output tag
TAG=${1:-q}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
timeout 400 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench.json
timeout 120 python scripts/jvrc_phase_profile.py 4096 > $OUT/phase.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d /tmp/pm -- python /root/repo/scripts/step_only.py 4096 4 > /tmp/pm.log 2>&1
python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_kernel<0, 1, 32" > $OUT/pmc_sq.csv
python -c "
import json
d=json.load(open('$OUT/bench.json')); r=d['roofline']
print('value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4), 'iso_ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), 'ovl_ms', round(r['overlapped']['avg_launch_ms'],4), {k:v for k,v in d['stepper_counters'].items() if k!='note'})
"
cat $OUT/phase.txt | tail -16
cat $OUT/pmc_sq.csv
