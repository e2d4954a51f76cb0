#!/bin/bash
# first GPU pass of round 3: primitive probe, GPU tests, bench + phase profile of the chain-solver kernel (with / without the dense path)
OUT=/root/repo/gpurun_out/r3a; mkdir -p $OUT
cd /root/repo
scripts/_bin/dpp_probe > $OUT/probe.txt 2>&1; cat $OUT/probe.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_default.json
python scripts/jvrc_phase_profile.py 4096 > $OUT/phase_default.txt 2>/dev/null
V=/root/repo/learninghumanoidwalking_amd/variants/liblhw_nodense.so
LHW_LIB=$V python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_nodense.json
LHW_LIB=$V python scripts/jvrc_phase_profile.py 4096 > $OUT/phase_nodense.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY"; do
  for L in default nodense; do
    rm -rf /tmp/pm
    if [ $L = nodense ]; then export LHW_LIB=$V; else unset LHW_LIB; fi
    timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/step_only.py 4096 4 > /tmp/pm.log 2>&1
    python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_kernel<0" > $OUT/pmc_sq_$L.csv
  done
done
unset LHW_LIB
python -c "
import json
for n in ('default','nodense'):
    d=json.load(open('$OUT/bench_%s.json'%n)); r=d['roofline']
    print(n, 'value', round(d['value']), 'sample', round(d['sample_s_per_iter'],4), 'opt', round(d['optimize_s_per_iter'],4), 'iso_ms', r['avg_launch_ms'], 'ovl_ms', r['overlapped']['avg_launch_ms'], d['stepper_counters'])
"
grep -E "ms/step|newton|kinematics|com/|crba|collision|constraints|velocity|smooth|euler" $OUT/phase_default.txt $OUT/phase_nodense.txt
cat $OUT/pmc_sq_default.csv $OUT/pmc_sq_nodense.csv
