#!/bin/bash
# usage: tools/exp_res.sh [-D...]: register / LDS / spill figures of ddc_rx_front for a set of build flags
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o /tmp/exp_res.s ua3reo-ddc-transceiver_amd/csrc/ddc_lib.hip 2>/dev/null
awk '/\.name:.*ddc_rx_front/{f=1} f&&/vgpr_count|sgpr_count|group_segment_fixed_size|spill_count|private_segment_fixed_size/{print} /\.name:/{if(f&&!/ddc_rx_front/)f=0}' /tmp/exp_res.s | sort -u
awk '/^_ZN3ddc12ddc_rx_frontE/{f=1} f&&/; Occupancy|; ScratchSize|; NumVgprs|; LDSByteSize/{print} /s_endpgm/{if(f)g=1} g&&/; Occupancy/{exit}' /tmp/exp_res.s | head
