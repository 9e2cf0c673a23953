#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per kernel.  usage: pmc_summary.py DIR..."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "copyBuffer" in k: continue
            print(d.split("/")[-1], k, {c: round(sum(v)/len(v),1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
