#!/bin/bash
# PMC counters of the update's GEMM kernel on its three big shapes (scripts/gemm_bench.py), one counter set per pass
OUT=/root/repo/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*\|TCP_[A-Z_]*STALL[A-Z_]*\|TCC_HIT_sum\|TCC_MISS_sum\|TCC_REQ_sum\|TCP_TCC_READ_REQ_sum" | sort -u > $OUT/names.txt
cat $OUT/names.txt | tr '\n' ' '
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pm -- python /root/repo/scripts/gemm_bench.py 32768 > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py /tmp/pm | grep -E "kernel,|gemm_f32_kernel" >> $OUT/gemm_pmc.csv
done
cat $OUT/gemm_pmc.csv | cut -c1-160
