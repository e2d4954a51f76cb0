#!/bin/bash
# round 5: cost of the in-wave policy step (variant that runs it twice), kernel-trace stats of the default bench command, PMC passes
# (SQ instruction mix / waits / HBM traffic, each in its own run) of the resident rollout kernel.  $1 = tag
TAG=${1:-r5_prof}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
V=/root/repo/learninghumanoidwalking_amd/variants/liblhw_pol2x.so
if [ -f $V ]; then
  for i in 1 2; do
    timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_default_$i.json
    LHW_LIB=$V timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_pol2x_$i.json
  done
fi
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $B --no-cpu-baseline > $OUT/kt.log 2>&1
cp /tmp/kt/*/*kernel_stats.csv $OUT/jvrc_walk_kernel_stats.csv
grep '^{' $OUT/kt.log | tail -1 > $OUT/bench_jvrc_walk_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python $B --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_rollout" > $OUT/jvrc_walk_rollout_pmc_$C.csv
done
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python $B --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_rollout" >> $OUT/jvrc_walk_rollout_pmc_sq.csv
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), "value", round(d["value"]), "sample", round(d["sample_s_per_iter"], 4), "opt", round(d["optimize_s_per_iter"], 4), d["roofline"].get("rollout_mode"))
PY
head -12 $OUT/jvrc_walk_kernel_stats.csv | cut -c1-160
cat $OUT/jvrc_walk_rollout_pmc_sq.csv $OUT/jvrc_walk_rollout_pmc_FETCH_SIZE.csv $OUT/jvrc_walk_rollout_pmc_WRITE_SIZE.csv
