#!/bin/bash
# usage: tools/fft_trace.sh <tag>: FFT parity tests, then kernel-trace statistics of the config-3 shaped bench
tag=${1:-fft}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
timeout -k 10 300 python -u -m pytest tests/test_audio_gpu.py tests/test_legacy_gpu.py tests/test_golden_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$tag/tests.log 2>&1 || { echo "tests failed"; tail -30 gpurun_out/$tag/tests.log; exit 1; }
tail -1 gpurun_out/$tag/tests.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/tr -o t -- python3 bench.py --workload ddc+fft --channels 256 --samples 4194304 --steps 10 --no-cpu-baseline > gpurun_out/$tag/log.txt 2>&1 || { echo "trace failed"; tail -5 gpurun_out/$tag/log.txt; exit 1; }
python3 tools/kstats.py gpurun_out/$tag/tr | grep -v synth
rm -rf gpurun_out/$tag/tr
