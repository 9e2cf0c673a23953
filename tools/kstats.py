#!/usr/bin/env python3
"""Prints per-kernel averages from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  %-36s calls %5s avg %8.2f us  min %8.2f max %8.2f" % (r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
