#!/bin/bash
# HBM-side traffic and SQ counters of the stepping task's resident rollout kernel (each --pmc set in its own run).
OUT=/root/repo/gpurun_out/step_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python $B --env jvrc_step --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_rollout" >> $OUT/jvrc_step_rollout_pmc.csv
done
cat $OUT/jvrc_step_rollout_pmc.csv
