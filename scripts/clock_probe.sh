#!/bin/bash
# sample shader clock / power while the rollout kernel runs (is the stepper power- or clock-limited?)
cd /root/repo
for V in default halfocc; do
  if [ $V = default ]; then unset LHW_LIB; else export LHW_LIB=/root/repo/learninghumanoidwalking_amd/variants/liblhw_$V.so; fi
  python scripts/rollout_only.py 4096 400 8 > /dev/null 2>&1 &
  PID=$!
  sleep 14
  for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo " [$V]"; sleep 1.5; done
  wait $PID
done
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo " [idle]"
