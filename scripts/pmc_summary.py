"""Summarise a rocprofv3 --pmc run: mean counter value per kernel (reads *_counter_collection.csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,dispatches,mean,total")
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    for c, v in cs.items():
        print(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.1f},{sum(v):.1f}")
