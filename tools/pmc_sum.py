"""Sums rocprofv3 counter_collection.csv files per (kernel, counter): python tools/pmc_sum.py DIR [kernel-substring]."""
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(float)
cnt = defaultdict(set)
sub = sys.argv[2] if len(sys.argv) > 2 else "blind_rotate"
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"].split("(")[0]].add(r["Dispatch_Id"])
for (k, c), v in sorted(tot.items()):
    print("%-40s %-28s %.6g  (%d dispatches)" % (k, c, v, len(cnt[k])))
