#!/usr/bin/env python3
"""Average per-dispatch counter values for this library's kernels from rocprofv3 --pmc csv output."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "ddc" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        v = sorted(v)
        print("   %-28s n=%d  max=%.4g  median=%.4g" % (c, len(v), v[-1], v[len(v) // 2]))
