"""From a rocprofv3 --pmc run with SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU (and SQ_ACTIVE_INST_VALU): per kernel, thread-cycles per VALU instruction
(a fully active wave64 instruction = 4 cycles x 64 lanes = 256) and issue cycles per instruction.  usage: pmc_lanes.py DIR"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    n = v.get("SQ_INSTS_VALU", 0)
    if n > 0:
        print(f"{k:60s} VALU {n:12.4g}  thread-cycles/inst {v.get('SQ_THREAD_CYCLES_VALU', 0) / n:7.1f}  (lanes at 4 cycles: {v.get('SQ_THREAD_CYCLES_VALU', 0) / n / 4:5.1f})  "
              f"issue cycles/inst {v.get('SQ_ACTIVE_INST_VALU', 0) / n:5.2f}")
