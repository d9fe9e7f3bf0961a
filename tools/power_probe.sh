#!/bin/bash
# usage: tools/power_probe.sh <tag> [bench args]: socket power and shader clock (rocm-smi) sampled while the bench
# runs a long timed region of the default step
tag=${1:-pwr}; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-copy-rate --no-secondary --no-check --steps 30000 "$@" > $out/bench.json 2> $out/bench.err &
bp=$!
sleep 7   # import, rows, warm-up
for i in $(seq 1 12); do
  timeout 10 rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|power|not supported" | grep -v "====" | tr -s ' \t' ' ' | tr '\n' '|' >> $out/samples.txt
  echo >> $out/samples.txt
  kill -0 $bp 2>/dev/null || break
done
wait $bp; echo "bench rc $?"
cat $out/samples.txt
python3 -c "
import json;d=json.load(open('$out/bench.json'));print('ms_per_step',d['ms_per_step'],'kernel',d['roofline']['avg_launch_ms'])"
