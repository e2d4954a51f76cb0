#!/bin/bash
# round 4: same-box A/B of the in-tree library against learninghumanoidwalking_amd/variants/liblhw_$1.so (default: head), then stepper parity
# usage: gpu_r4_ab.sh VARIANT OUTDIR [reps of jvrc_walk] [other envs...]
cd "$(dirname "$0")/.."
V=${1:-head}; O=gpurun_out/${2:-r4ab}; R=${3:-4}; shift 3; E=${@:-jvrc_step h1_walk}; mkdir -p $O
{
for rep in $(seq 1 $R); do
  unset LHW_LIB; timeout 120 python scripts/step_time.py 4096 2>/dev/null | tail -1
  LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/liblhw_$V.so timeout 120 python scripts/step_time.py 4096 2>/dev/null | tail -1
done
for env in $E; do
  unset LHW_LIB; timeout 120 python scripts/step_time.py 4096 $env 2>/dev/null | tail -1
  LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/liblhw_$V.so timeout 120 python scripts/step_time.py 4096 $env 2>/dev/null | tail -1
done
} > $O/ab.txt 2>&1
unset LHW_LIB
timeout 900 python -m pytest tests/test_jvrc_gpu.py tests/test_h1_gpu.py tests/test_h1_walk_gpu.py tests/test_jvrc_step_gpu.py tests/test_model_variants_gpu.py tests/test_fullsize_gpu.py tests/test_freerun_gpu.py -m gpu -q 2>&1 | tail -5 > $O/parity.txt
cat $O/ab.txt; cat $O/parity.txt
