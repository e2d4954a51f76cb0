#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6u
V=/root/repo/learninghumanoidwalking_amd/variants
for L in "" nostore nomfma neither; do
  echo "variant ${L:-product}"; ( [ -n "$L" ] && export LHW_LIB=$V/liblhw_$L.so; timeout 100 python scripts/strip_bench.py 32768; timeout 100 python scripts/strip_bench.py 65536 ) 2>/dev/null | grep " strip"
done | tee gpurun_out/r6u/strip_ablation.txt
