#!/bin/bash
# usage: [BENCH_ARGS="..."] tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# one rocprofv3 --pmc pass over a short bench run
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle $BENCH_ARGS > gpurun_out/pmc_$tag/run.log 2>&1
echo "pmc $tag exit=$?"
