#!/bin/bash
# usage: tools/ab_lib.sh <tag> <other .so> [bench args]: the default bench shape with the library of the tree
# against another build of it (UA3REO_DDC_LIB), alternating, on one box.
set -o pipefail
tag=${1:-ablib}; other=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export UA3REO_DDC_LIB=$other; else unset UA3REO_DDC_LIB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-copy-rate --no-secondary --steps 40 "$@" > $out/${which}_$rep.json 2> $out/${which}_$rep.err || { echo "$which failed"; tail -5 $out/${which}_$rep.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$out/${which}_$rep.json'));print('$which',$rep,'ms_per_step',d['ms_per_step'],'kernel',d['roofline']['avg_launch_ms'],'first_window',d['config']['clock_settle']['first_window_ms_per_step'])"
  done
done
