#!/usr/bin/env python3
"""Sum rocprofv3 --pmc csv output per (kernel, counter): mean over dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"][:52], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
for (k, c), v in sorted(acc.items()):
    if sys.argv[2] in k:
        print("%-54s %-32s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
