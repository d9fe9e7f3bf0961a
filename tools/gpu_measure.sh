#!/bin/bash
# usage: tools/gpu_measure.sh <tag>
# One measurement round on the GPU box: default bench line, kernel-trace stats of
# the same command, two PMC passes (HBM read / write bytes), and the config-3 and
# config-4 shaped workloads.  Every GPU step has its own time limit; the script
# stops at the first failure.
set -o pipefail
tag=${1:-run}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
step() { echo "== $*"; }

step bench
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -20 $out/bench.err; exit 1; }
cat $out/bench.json

step kernel-trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-secondary > $out/trace.log 2>&1 || { echo "trace failed"; tail -20 $out/trace.log; exit 1; }
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -r cat > $out/kernel_stats.csv
head -20 $out/kernel_stats.csv
# the same trace by phase: the bench runs 3 warm-up, 20 first-window, the settle calls and then the 20 timed steps
python3 - "$(find $out/trace -name "*kernel_trace.csv" | head -1)" > $out/trace_phases.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ddc_rx_front" in r["Kernel_Name"] or "ddc_rx_tail" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ("ddc_rx_front", "ddc_rx_tail"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    full = [x for x in d if x > 0.5 * max(d)]  # without the small parity-check call
    print("%-18s launches %3d | first 23 (warm-up + first window): avg %7.1f us | last 20 (the timed region): avg %7.1f min %7.1f max %7.1f us"
          % (name, len(full), sum(full[:23]) / 23, sum(full[-20:]) / 20, min(full[-20:]), max(full[-20:])))
PY
cat $out/trace_phases.txt

step pmc-fetch
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle > $out/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -20 $out/pmc_fetch.log; exit 1; }
step pmc-write
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle > $out/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -20 $out/pmc_write.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc_fetch > $out/pmc_summary.txt
python3 tools/pmc_summary.py $out/pmc_write >> $out/pmc_summary.txt
cat $out/pmc_summary.txt
# keep only the summaries (the raw csv files are large)
rm -rf $out/pmc_fetch $out/pmc_write $out/trace

step config-3-shape
timeout -k 10 300 python3 bench.py --workload ddc+fft --channels 256 --samples 4194304 --steps 10 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err || { echo "cfg3 failed"; tail -20 $out/bench_cfg3.err; exit 1; }
cat $out/bench_cfg3.json
step config-4-shape
timeout -k 10 300 python3 bench.py --workload tx --channels 256 --samples 4194304 --steps 10 > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { echo "cfg4 failed"; tail -20 $out/bench_cfg4.err; exit 1; }
cat $out/bench_cfg4.json
step config-4-pmc
BENCH_ARGS="--workload tx --channels 256 --samples 4194304" bash tools/pmc_pass.sh txw WRITE_SIZE && BENCH_ARGS="--workload tx --channels 256 --samples 4194304" bash tools/pmc_pass.sh txf FETCH_SIZE || { echo "tx pmc failed"; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_txw > $out/pmc_summary_tx.txt
python3 tools/pmc_summary.py gpurun_out/pmc_txf >> $out/pmc_summary_tx.txt
rm -rf gpurun_out/pmc_txw gpurun_out/pmc_txf
cat $out/pmc_summary_tx.txt
echo "== done"
