#!/bin/bash
# Instruction-cache counters of the resident rollout kernel (the sub-step is ~30 k instructions of straight-line code, ~180 KB,
# streamed once per sub-step by every wave).  usage: gpu_icache_pmc.sh [env]
ENVN=${1:-jvrc_walk}
mkdir -p /root/repo/gpurun_out/icache_pmc
CSV=/root/repo/gpurun_out/icache_pmc/${ENVN}_icache.csv
: > "$CSV"
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  D=$(mktemp -d /tmp/pm.XXXXXX)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d "$D" -- python $B --env $ENVN --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python /root/repo/scripts/pmc_summary.py "$D" | grep -E "kernel,|humanoid_rollout" >> "$CSV"
done
cat "$CSV"
