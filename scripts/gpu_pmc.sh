#!/bin/bash
# Counter passes of the resident rollout kernel, each --pmc set in its own run (run on the GPU box via gpurun).
# usage: gpu_pmc.sh SET [env]     SET = traffic | icache | wait | tcp      env = jvrc_walk (default) | jvrc_step | h1 | h1_walk
#   traffic: FETCH_SIZE / WRITE_SIZE + the SQ instruction / cycle / fp64 counters (what scripts/collect_profiles.sh collects for jvrc_walk)
#   icache : instruction-cache requests / hits / misses, instruction requests to the L2
#   wait   : outstanding-instruction integrals (SQ_INST_LEVEL_*) against the instruction counts
#   tcp    : vector-L1 accesses, L2 read / write requests and read latency
SET=${1:-traffic}; ENVN=${2:-jvrc_walk}
mkdir -p /root/repo/gpurun_out/pmc
CSV=/root/repo/gpurun_out/pmc/${ENVN}_${SET}.csv
: > "$CSV"
cd /tmp && export TMPDIR=/tmp
case $SET in
  traffic) SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU");;
  icache)  SETS=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY");;
  wait)    SETS=("SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAVE_CYCLES");;
  tcp)     SETS=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum");;
  *) echo "unknown set $SET"; exit 2;;
esac
NE=4096; case $ENVN in h1|h1_walk) NE=8192;; esac
for C in "${SETS[@]}"; do
  D=$(mktemp -d /tmp/pm.XXXXXX)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d "$D" -- python /root/repo/bench.py --env $ENVN --num-envs $NE --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
  python /root/repo/scripts/pmc_summary.py "$D" | grep -E "kernel,|humanoid_rollout" >> "$CSV"
done
cat "$CSV"
