#!/usr/bin/env python3
"""Dev tool: average of every counter per kernel (name filter) out of rocprofv3 --pmc counter_collection CSVs.
usage: pmc_fold.py <name filter> <csv> [<csv> ...]"""
import collections, csv, sys

flt = sys.argv[1]
for path in sys.argv[2:]:
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if flt not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(tot):
        print("%-34s %16.1f per launch (%d launches)" % (k, tot[k] / max(1, len(n[k])), len(n[k])))
