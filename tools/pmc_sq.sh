#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [bench args]
# Scheduler / LDS counter passes over a short bench run of the shipped ingest kernel,
# each pass a run of its own (at most 8 SQ counters a pass), summary into
# gpurun_out/<tag>/pmc_sq.txt.  Stops at the first pass that fails.
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
pass() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-copy-rate --no-secondary --no-settle $BENCH_ARGS > $out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
    python3 tools/pmc_summary.py $out/pmc_$name >> $out/pmc_sq.txt
    rm -rf $out/pmc_$name
}
: > $out/pmc_sq.txt
pass issue SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass mfma SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CU_CYCLES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
cat $out/pmc_sq.txt
