#!/bin/bash
# usage: tools/exp_insts.sh <tag> <name> [<name> ...]: SQ_INSTS_VALU per ddc_rx_front launch for experimental builds
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for n in "$@"; do
    export UA3REO_DDC_LIB=$PWD/build/exp/lib_$n.so
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag/pmc_$n -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle > gpurun_out/$tag/pmc_$n.log 2>&1 || { echo "$n failed"; tail -5 gpurun_out/$tag/pmc_$n.log; exit 1; }
    echo "== $n"; python3 tools/pmc_summary.py gpurun_out/$tag/pmc_$n | grep -A2 "ddc_rx_front"
    rm -rf gpurun_out/$tag/pmc_$n
done
