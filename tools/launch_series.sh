#!/bin/bash
# usage: tools/launch_series.sh <tag> [bench args]: the duration of every ddc_rx_front_cic / ddc_tx_duc launch of one bench
# run in launch order (rocprofv3 --kernel-trace), to see how the clock the chip holds moves over a run
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/trace -o t -- python3 bench.py --no-cpu-baseline --no-secondary --no-check --no-settle "$@" > gpurun_out/$tag/trace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/$tag/trace.log; exit 1; }
f=$(find gpurun_out/$tag/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_series.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ddc_rx_front" in r["Kernel_Name"] or "ddc_tx_duc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  %8.1f us" % ((s - t0) / 1e6, (e - s) / 1e3))
PY
rm -rf gpurun_out/$tag/trace
cat gpurun_out/${tag}_series.txt
