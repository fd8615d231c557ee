#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.  usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel substring]"""
import csv, glob, json, sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if want and want not in name:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name].add(r["Dispatch_Id"])
out = {k: {"launches": len(launches[k]), **{c: v / max(1, len(launches[k])) for c, v in sorted(acc[k].items())}} for k in acc}
print(json.dumps(out, indent=1))
