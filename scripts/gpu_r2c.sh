#!/bin/bash
# round-2 GPU pass C: grouping of the rollout, instruction-mix counters of the two-envs-per-wave kernel
set -u
OUT=/root/repo/gpurun_out/r2f
mkdir -p $OUT
cd /root/repo
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for G in 1 2; do
  LHW_ROLLOUT_GROUPS=$G python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/bench_g$G.err | tail -1 > $OUT/bench_walk_g$G.json
done
LHW_ROLLOUT_GROUPS=1 python bench.py --env h1 --num-envs 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_h1_g1.json
python scripts/jvrc_phase_profile.py 4096 > $OUT/phase_fast.txt 2>&1
python scripts/jvrc_phase_profile.py 2048 > $OUT/phase_fast_2048.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/step_only.py 4096 4 > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_kernel<0" >> $OUT/walk_step_pmc_sq.csv
done
cd /root/repo
python - <<'PY' > $OUT/summary.txt
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r2f/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f'%d['value'], 'sample_s %.3f opt_s %.3f'%(d['sample_s_per_iter'], d['optimize_s_per_iter']), 'launch_ms %.3f wall_ms/step %.3f'%(r['avg_launch_ms'], r['aggregate']['wall_ms_per_control_step']))
    except Exception as e: print(f, 'ERR', e)
PY
cat $OUT/summary.txt; head -16 $OUT/phase_fast.txt; head -8 $OUT/phase_fast_2048.txt; cat $OUT/walk_step_pmc_sq.csv
