#!/bin/bash
# round 4: A/B of stepper build variants (scripts/gpu_variants.sh) + parity of one candidate
cd "$(dirname "$0")/.."
O=gpurun_out/r4b; mkdir -p $O
bash scripts/gpu_variants.sh > $O/variants.txt 2>&1
C=${1:-liblhw_d_fresh_nolicm.so}
LHW_LIB=$PWD/learninghumanoidwalking_amd/variants/$C timeout 900 python -m pytest tests/test_jvrc_gpu.py tests/test_h1_gpu.py tests/test_h1_walk_gpu.py tests/test_jvrc_step_gpu.py tests/test_model_variants_gpu.py tests/test_fullsize_gpu.py tests/test_freerun_gpu.py -m gpu -q 2>&1 | tail -5 > $O/parity_candidate.txt
cat $O/variants.txt; cat $O/parity_candidate.txt
