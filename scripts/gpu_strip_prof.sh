#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; LHW_MLP_STRIP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
python - <<'P'
import csv,glob
f=glob.glob('/tmp/kt/*/*kernel_stats.csv')[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<14: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'], 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
P
