#!/bin/bash
# What the stepper's waves wait for: time integrals of outstanding VMEM / LDS / SMEM instructions (SQ_INST_LEVEL_*) against their counts.
ENVN=${1:-jvrc_walk}
mkdir -p /root/repo/gpurun_out/wait_pmc
CSV=/root/repo/gpurun_out/wait_pmc/${ENVN}_wait.csv
: > "$CSV"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "Counter_Name" | grep -E "SQ_INST_LEVEL|SQ_WAIT|SQ_WAVE_DEP|SQ_INSTS_VMEM|SQ_INSTS_FLAT|SQ_LEVEL_WAVES|SQ_INSTS_SMEM|SQ_INSTS_LDS|SQ_ACTIVE_INST_(VMEM|SCA|MISC|FLAT)" | sed 's/.*:\s*//' | tr '\n' ' ' > /root/repo/gpurun_out/wait_pmc/avail.txt
B=/root/repo/bench.py
for C in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  D=$(mktemp -d /tmp/pm.XXXXXX)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d "$D" -- python $B --env $ENVN --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
  python /root/repo/scripts/pmc_summary.py "$D" | grep -E "humanoid_rollout" >> "$CSV"
done
cat /root/repo/gpurun_out/wait_pmc/avail.txt; echo; cat "$CSV"
