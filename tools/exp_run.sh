#!/bin/bash
# usage: tools/exp_run.sh <tag> <name> [<name> ...]: bench line (parity check on, no CPU leg) of each experimental build
tag=$1; shift
mkdir -p gpurun_out/$tag
for n in "$@"; do
    UA3REO_DDC_LIB=$PWD/build/exp/lib_$n.so timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-copy-rate --no-secondary --steps 20 --warmup 3 $BENCH_EXTRA > gpurun_out/$tag/$n.json 2> gpurun_out/$tag/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/$tag/$n.err; exit 1; }
    python3 - gpurun_out/$tag/$n.json $n <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("%-16s value %9.0f MS/s  step %.4f ms  kernel %.4f ms  frac %.4f  rest %.4f  check: %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["roofline"].get("rest_of_call_ms", 0.0), r["config"]["parity_check"][:20]))
PY
done
