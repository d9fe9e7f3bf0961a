#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [bench args]: vector instructions and cycles per launch of the
# ingest kernel (one short counter pass) next to a plain bench line of the same build
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle "$@" > $out/pmc.log 2>&1 || { echo "pass failed"; tail -5 $out/pmc.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc > $out/pmc_quick.txt
rm -rf $out/pmc
grep -A7 "ddc_rx_front" $out/pmc_quick.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --no-copy-rate "$@" > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('$out/bench.json'));print('value',d['value'],'ms_per_step',d['ms_per_step'],'frac',d['roofline']['frac'],'kernel_ms',d['roofline']['avg_launch_ms'],'first_window',d['config']['clock_settle']['first_window_ms_per_step'])"
