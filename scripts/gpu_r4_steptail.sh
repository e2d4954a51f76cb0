#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4e; mkdir -p $O
timeout 300 python scripts/step_tail_trace.py > $O/step_trace.txt 2>&1; cat $O/step_trace.txt
