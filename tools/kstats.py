#!/usr/bin/env python3
"""Library-kernel lines (name, calls, average / min / max ns) of a rocprofv3 kernel_stats.csv found under a directory."""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Name"].split("(")[0]
        if "ddc::" in n:
            print("%-30s calls=%-4s avg=%9.0f min=%8s max=%8s" % (n.replace("ddc::", ""), r["Calls"], float(r["AverageNs"]),
                                                                 r["MinNs"], r["MaxNs"]))
