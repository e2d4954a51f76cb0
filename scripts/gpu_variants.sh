#!/bin/bash
# time stepper variants (learninghumanoidwalking_amd/variants/liblhw_*.so) with the phase profiler, same box
OUT=/root/repo/gpurun_out/var
mkdir -p $OUT
cd /root/repo
for rep in 1 2; do
for f in learninghumanoidwalking_amd/variants/liblhw_*.so; do
  v=$(basename $f .so)
  LHW_LIB=/root/repo/$f python scripts/jvrc_phase_profile.py 4096 > $OUT/${v}.txt 2>&1
  echo "$v $(grep 'ms/step' $OUT/${v}.txt)"
  if [ $rep = 1 ]; then grep -E "kinematics|com/|crba|velocity|newton|collision|constraints|euler|detail" $OUT/${v}.txt | tr '\n' ' ' | sed 's/cyc\/substep//g; s/  */ /g'; echo; fi
done
done
