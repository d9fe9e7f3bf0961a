#!/bin/bash
# usage: tools/exp_build.sh <name> [-D...]: one experimental build of the library
# into build/exp/lib_<name>.so (selected at run time with UA3REO_DDC_LIB)
name=$1; shift
mkdir -p build/exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o build/exp/lib_$name.so \
    ua3reo-ddc-transceiver_amd/csrc/ddc_lib.hip ua3reo-ddc-transceiver_amd/csrc/ddc_legacy.cpp 2>&1 | grep -E "error|spill" | head
ls -la build/exp/lib_$name.so
