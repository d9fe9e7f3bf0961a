export TMPDIR=/tmp
out=gpurun_out/mxw; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle > $out/pmc.log 2>&1 || { echo fail; tail -5 $out/pmc.log; exit 1; }
python3 tools/pmc_summary.py $out/pmc | grep -A9 "front_mx"
