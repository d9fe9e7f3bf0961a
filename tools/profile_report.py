#!/usr/bin/env python3
"""Condenses one tools/gpu_measure.sh output directory (gpurun_out/<tag>) into the
text summary kept under profiles/:  python tools/profile_report.py <tag> <out.txt>"""
import csv
import sys

tag, out_path = sys.argv[1], sys.argv[2]
d = "gpurun_out/" + tag
rows = list(csv.DictReader(open(d + "/kernel_stats.csv")))
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (MI355X; default workload: 1024 channels x "
       "2^20 samples: 3 warm-up + 20 first-window + the settle calls + 20 timed steps, and 1 parity-check call of 131072 samples)",
       "# library kernels only (ddc_synth_* generate the synthetic ADC rows before the timed region)",
       "%-28s %6s %14s %12s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns")]
for r in rows:
    n = r["Name"].split("(")[0]
    if "ddc::" in n:
        out.append("%-28s %6s %14s %12.0f %10s %10s" % (n.replace("ddc::", ""), r["Calls"], r["TotalDurationNs"],
                                                        float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
out.append("")
out.append("# the parity-check call is the MinNs outlier of every kernel; full-size launches only:")
for r in rows:
    n = r["Name"].split("(")[0]
    if "ddc_rx_front" in n:
        c = int(r["Calls"])
        out.append("#   %s full-size average = %.1f us" % (n.replace("ddc::", ""),
            (float(r["TotalDurationNs"]) - float(r["MinNs"])) / (c - 1) / 1e3))
try:
    out.append("")
    out.append("# the same trace split by the bench's phases (the timed region = the last 20 full-size launches):")
    out.append(open(d + "/trace_phases.txt").read().rstrip())
except OSError:
    pass
out.append("")
out.append("# bench.py line of the same build (HIP events on the ctx stream):")
out.append(open(d + "/bench.json").read().strip())
out.append("")
out.append("# PMC passes (one counter per run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 3 "
           "--warmup 1), KiB per dispatch")
out.append(open(d + "/pmc_summary.txt").read().rstrip())
out.append("")
out.append("# other workload shapes, same build")
for f in ("bench_cfg3.json", "bench_cfg4.json"):
    try:
        out.append(open(d + "/" + f).read().strip())
    except OSError:
        pass
try:
    out.append("")
    out.append("# PMC passes on the TX shape (bench.py --workload tx --channels 256 --samples 4194304), KiB per dispatch")
    out.append(open(d + "/pmc_summary_tx.txt").read().rstrip())
except OSError:
    pass
open(out_path, "w").write("\n".join(out) + "\n")
print("wrote", out_path)
