#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/strip_bench.py 65536 > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|strip"
done
