#!/bin/bash
# usage: tools/trace_stats.sh <tag> [bench args]: rocprofv3 kernel-trace statistics of one bench run,
# library kernels only, into gpurun_out/<tag>_stats.txt
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -o t -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag/trace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/$tag/trace.log; exit 1; }
f=$(find gpurun_out/$tag/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_stats.txt <<'PY'
import csv, sys
print("%-28s %6s %12s %10s %10s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns"))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if "ddc::" in n:
        print("%-28s %6s %12.0f %10s %10s" % (n.replace("ddc::", ""), r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
rm -rf gpurun_out/$tag/trace
cat gpurun_out/${tag}_stats.txt
