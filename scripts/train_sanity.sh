#!/bin/bash
# End-to-end training sanity runs on the GPU box (reward / episode-length trends of every env, LSTM path timing).
# Outputs under gpurun_out/train_raw/.
set -u
OUT=/root/repo/gpurun_out/train_raw
mkdir -p $OUT
cd /root/repo
F='Iteration|Mean Eprew|Mean Eplen|fps=|Sampling took|Optimizer took'
for E in jvrc_step h1 h1_walk; do
  rm -rf /tmp/tl_$E
  timeout 400 python run_experiment.py train --env $E --num-envs 4096 --minibatch-size 32768 --n-itr 30 --eval-freq 1000 --logdir /tmp/tl_$E --seed 0 2>&1 | grep -E "$F" > $OUT/train_${E}_30iters.log
done
rm -rf /tmp/tl_rec
timeout 500 python run_experiment.py train --env jvrc_walk --recurrent --num-envs 2048 --max-traj-len 200 --minibatch-size 512 --n-itr 4 --eval-freq 1000 --logdir /tmp/tl_rec --seed 0 2>&1 | grep -E "$F|rror" > $OUT/train_jvrc_walk_recurrent_4iters.log
ls -la $OUT
