#!/bin/bash
# usage: tools/audio_trace.sh <tag>: kernel-trace statistics of tools/audio_timing.py and tools/cfg2_timing.py
tag=${1:-audio}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for w in "audio_timing.py 1024 16" "cfg2_timing.py"; do
  set -- $w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/tr -o t -- python3 tools/$@ > gpurun_out/$tag/log.txt 2>&1 || { echo "trace failed"; tail -5 gpurun_out/$tag/log.txt; exit 1; }
  echo "== $w"; grep -v amdgpu.ids gpurun_out/$tag/log.txt | tail -30
  python3 tools/kstats.py gpurun_out/$tag/tr | grep -i "audio\|fft"
  rm -rf gpurun_out/$tag/tr
done
