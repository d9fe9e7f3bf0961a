#!/bin/bash
# usage: tools/clock_probe.sh <tag> [bench args]: effective shader clock of the library's kernels =
# GRBM_GUI_ACTIVE / 8 / duration, from one rocprofv3 pass with counters and kernel trace
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/clk_$tag
mkdir -p $out
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle "$@" > $out/run.log 2>&1 || { echo "probe failed"; tail -5 $out/run.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = collections.defaultdict(list); cyc = collections.defaultdict(list)
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k in dur:
    if "ddc" in k and k in cyc:
        a = sorted(dur[k]); b = sorted(cyc[k])
        t = a[len(a) // 2]; c = b[len(b) // 2] / 8
        if t > 100000:
            print("%-28s median %8.1f us  %9.0f cycles  %.3f GHz" % (k.replace("ddc::", ""), t / 1e3, c, c / t))
PY
