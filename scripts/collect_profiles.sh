#!/bin/bash
# Run on the GPU box (via gpurun): bench lines of every env, kernel-trace stats of the default bench command, HBM traffic and SQ
# counter passes of the resident rollout kernel (each --pmc set in its own run), per-phase cycle breakdown, wave-time spread, the
# update's GEMM / strip micro-benchmarks and a training log.  Outputs under gpurun_out/profiles_raw/ (copy the summaries into
# profiles/ with the round prefix afterwards).  QUICK=1 skips the counter passes and the training run.
set -u
OUT=/root/repo/gpurun_out/profiles_raw
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
timeout 300 python $B --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_1gpu.json
timeout 200 python $B --steps 5 --warmup 2 --no-cpu-baseline --task-hook walking 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_1gpu_task_plugin.json
timeout 200 python $B --env h1 --num-envs 8192 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_h1_8192_1gpu.json
timeout 100 python $B --env cartpole --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_cartpole_1gpu.json
timeout 200 python $B --env jvrc_step --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_jvrc_step_1gpu.json
timeout 200 python $B --env h1_walk --num-envs 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_h1_walk_8192_1gpu.json
timeout 200 python $B --env h1 --num-envs 8192 --fp16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_h1_8192_fp16_1gpu.json
LHW_ROLLOUT_MODE=steps timeout 200 python $B --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_jvrc_walk_1gpu_launch_per_step.json
# per-kernel durations of the default bench command (the rollout kernel's average must agree with the bench line's HIP events)
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $B --no-cpu-baseline > $OUT/kt.log 2>&1
cp /tmp/kt/*/*kernel_stats.csv $OUT/jvrc_walk_kernel_stats.csv
grep '^{' $OUT/kt.log | tail -1 > $OUT/bench_jvrc_walk_under_rocprof.json
for E in h1 jvrc_step h1_walk; do
  NE=4096; [ $E != jvrc_step ] && NE=8192
  rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $B --env $E --num-envs $NE --steps 2 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
  cp /tmp/kt/*/*kernel_stats.csv $OUT/${E}_kernel_stats.csv
done
timeout 120 python /root/repo/scripts/jvrc_phase_profile.py 4096 > $OUT/jvrc_walk_phase_cycles.txt 2>/dev/null
( timeout 100 python /root/repo/scripts/rollout_wave_spread.py jvrc_walk 4096 3; timeout 100 python /root/repo/scripts/rollout_wave_spread.py jvrc_step 4096 1; timeout 100 python /root/repo/scripts/rollout_wave_spread.py h1 8192 1; timeout 100 python /root/repo/scripts/rollout_wave_spread.py h1_walk 8192 1 ) 2>/dev/null | grep -v "^Using" > $OUT/rollout_wave_spread.txt
timeout 200 python /root/repo/scripts/gemm_bench.py 32768 > $OUT/ppo_gemm_shapes.txt 2>/dev/null
( timeout 100 python /root/repo/scripts/strip_bench.py 65536; timeout 100 python /root/repo/scripts/strip_bench.py 32768 ) 2>/dev/null | grep rows > $OUT/ppo_strip_bench.txt
if [ "${QUICK:-0}" != 1 ]; then
# counter passes of the resident rollout kernels (scripts/gpu_pmc.sh: every --pmc set in its own rocprofv3 run; bench.py reads
# profiles/r06_<env>_rollout_pmc.csv): the headline env, config 5's stepper (h1 @ 8192) and the stepping task
for E in jvrc_walk h1 jvrc_step; do
  bash /root/repo/scripts/gpu_pmc.sh traffic $E > /tmp/pmc_$E.log 2>&1
  cp /root/repo/gpurun_out/pmc/${E}_traffic.csv $OUT/${E}_rollout_pmc.csv
done
# end-to-end sanity: 100 PPO iterations of jvrc_walk on the stand-in robot (reward / episode length trend)
rm -rf /tmp/train_log; timeout 600 python /root/repo/run_experiment.py train --env jvrc_walk --num-envs 4096 --minibatch-size 32768 --n-itr 100 --eval-freq 1000 --logdir /tmp/train_log --seed 0 2>&1 | grep -E "Iteration|Mean Eprew|Mean Eplen|fps=|Sampling took|Optimizer took" > $OUT/train_jvrc_walk_100iters.log
fi
ls $OUT
