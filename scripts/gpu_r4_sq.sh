#!/bin/bash
# SQ counter passes of the whole-batch control-step launch (the non-QUICK part of collect_profiles.sh) -> gpurun_out/r4sq/jvrc_walk_step_pmc_sq.csv
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r4sq; mkdir -p $O; rm -f $O/jvrc_walk_step_pmc_sq.csv
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/step_only.py 4096 4 > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|humanoid_kernel<0" >> $O/jvrc_walk_step_pmc_sq.csv
done
grep "0, 1, 32" $O/jvrc_walk_step_pmc_sq.csv | awk -F, '{print $(NF-3), $(NF-1)}' | head -40
