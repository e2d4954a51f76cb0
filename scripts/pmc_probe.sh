#!/bin/bash
# list the counters rocprofv3 offers on this box and collect latency / busy ones for the control-step kernel
OUT=/root/repo/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -o "Name:[[:space:]]*[A-Za-z_0-9]*" $OUT/counters_all.txt | awk '{print $NF}' | sort -u > $OUT/names.txt
wc -l $OUT/names.txt
grep -E "LEVEL|LATENCY|TA_BUSY|TCP_.*BUSY|TD_.*BUSY|LDS.*BUSY|SQ_INSTS_VALU_|SQ_VALU|SQ_THREAD_CYCLES|SQ_INST_CYCLES|ACCUM|SQ_WAVE32|SQ_LDS_|SQ_ACTIVE_INST|SQ_WAIT_INST|SQ_EXP|SQ_INSTS_SALU|SQ_WAIT" $OUT/names.txt | tr '\n' ' '
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/step_only.py 4096 4 > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "humanoid_kernel<0, 1, 32" | cut -c60-140 >> $OUT/step_probe.csv
done
cat $OUT/step_probe.csv
