#!/bin/bash
# usage: [PLANS="auto valu split"] tools/ab_plan.sh <tag> [bench args]: the default bench shape with the RX call plans
# (auto: two kernels, moments on the matrix cores; valu: two kernels, vector-ALU ingest; split: six kernels),
# alternating, on one box.
set -o pipefail
tag=${1:-ab}; shift
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2 3; do
  for plan in ${PLANS:-auto valu split}; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-copy-rate --no-secondary --steps 40 --rx-plan $plan "$@" > $out/${plan}_$rep.json 2> $out/${plan}_$rep.err || { echo "$plan failed"; tail -5 $out/${plan}_$rep.err; exit 1; }
    python3 -c "
import json,sys;d=json.load(open('$out/${plan}_$rep.json'));print('$plan',$rep,'ms_per_step',d['ms_per_step'],'kernel',d['roofline']['avg_launch_ms'],'rest',d['roofline']['rest_of_call_ms'],'first_window',d['config']['clock_settle']['first_window_ms_per_step'])"
  done
done
