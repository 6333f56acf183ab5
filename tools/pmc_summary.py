#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ipm::", "")[:60]
            acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for nm, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            print("  %-62s" % nm, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
