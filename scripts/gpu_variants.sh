#!/bin/bash
# time stepper variants (learninghumanoidwalking_amd/variants/liblhw_*.so) against the in-tree library, same box, interleaved
cd /root/repo
for rep in 1 2 3; do
  unset LHW_LIB; timeout 120 python scripts/step_time.py 4096 2>/dev/null | tail -1
  for f in learninghumanoidwalking_amd/variants/liblhw_*.so; do
    LHW_LIB=/root/repo/$f timeout 120 python scripts/step_time.py 4096 2>/dev/null | tail -1
  done
done
