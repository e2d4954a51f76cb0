#!/bin/bash
# round 4: the data-parallel mismatch of GPUTEST_r03 -- full suite on the fixed kernels, the poison demonstration (round-3 Hessian
# loop vs the fix, both with LDS / allocations 0xFF-filled), the suite under poison, then the two-rank test repeated.
cd "$(dirname "$0")/.."
O=gpurun_out/r4a; mkdir -p $O
V=learninghumanoidwalking_amd/variants
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_full.txt
echo "--- r3 Hessian loop + poison" > $O/poison_demo.txt
LHW_LIB=$PWD/$V/liblhw_poison_r3.so LHW_POISON=1 timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k two_ranks 2>&1 | tail -25 >> $O/poison_demo.txt
echo "--- fixed + poison" >> $O/poison_demo.txt
LHW_LIB=$PWD/$V/liblhw_poison.so LHW_POISON=1 timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k two_ranks 2>&1 | tail -25 >> $O/poison_demo.txt
LHW_LIB=$PWD/$V/liblhw_poison.so LHW_POISON=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_poison.txt
for i in $(seq 1 20); do
  timeout 300 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k two_ranks 2>&1 | tail -1
done > $O/dp_repeat.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_full.txt; cat $O/poison_demo.txt | grep -E "passed|failed|---|Error|differs"; tail -3 $O/pytest_poison.txt; sort $O/dp_repeat.txt | uniq -c; cat $O/bench.json | cut -c1-600
