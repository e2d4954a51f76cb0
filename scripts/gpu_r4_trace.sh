#!/bin/bash
# per-dispatch kernel trace of a short bench run -> gpurun_out/r4trace/kernel_trace.csv (idle-gap analysis of the update phase)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r4trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/run.log 2>&1
ls -la /tmp/kt/*/ >> $O/run.log
cp /tmp/kt/*/*kernel_trace.csv $O/kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_trace.csv")))
print(len(rows), list(rows[0].keys()))
PY
