"""Python view of the C ABI in include/ua3reo_ddc.h (ctypes, no torch types in
the signatures; torch is only used by callers to own device memory)."""
from .capi import (Ddc, DdcError, lib, lib_path, build_native, pack_adc12, unpack_adc12, synth_adc,
                   phrase_from_frequency, frequency_from_phrase, pack_rx_frames, unpack_tx_frames,
                   parse_params, parse_status, DECIMATION, MODES, AudioParams, AudioStatus, audio_params, AutoGain, debug_table, synth_adc_dev, shard_range, comm_unique_id)

__all__ = ["Ddc", "DdcError", "lib", "lib_path", "build_native", "pack_adc12", "unpack_adc12",
           "synth_adc", "phrase_from_frequency", "frequency_from_phrase", "pack_rx_frames",
           "unpack_tx_frames", "parse_params", "parse_status", "DECIMATION", "MODES", "AudioParams", "AudioStatus",
           "audio_params", "AutoGain", "debug_table", "synth_adc_dev", "shard_range", "comm_unique_id"]
