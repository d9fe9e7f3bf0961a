"""ctypes view of the firmware-style globals exported by the library
(include/ua3reo_legacy.h), for the tests."""
import ctypes as C

import numpy as np


class VFO(C.Structure):
    _fields_ = [("Freq", C.c_uint32), ("Mode", C.c_uint8), ("Filter_Width", C.c_uint16)]


class TrxSettings(C.Structure):
    _fields_ = [("current_vfo", C.c_bool), ("VFO_A", VFO), ("VFO_B", VFO), ("AGC", C.c_bool), ("Preamp", C.c_bool),
                ("DNR", C.c_bool), ("Agc_speed", C.c_uint8), ("Volume", C.c_uint8), ("Mute", C.c_bool),
                ("RF_Power", C.c_uint8), ("FM_SQL_threshold", C.c_uint8), ("RF_Gain", C.c_uint8),
                ("FFT_Zoom", C.c_uint8), ("NotchFilter", C.c_bool), ("NotchFC", C.c_uint16), ("CWDecoder", C.c_bool),
                ("FFT_Enabled", C.c_bool), ("CW_GENERATOR_SHIFT_HZ", C.c_uint16), ("FFT_Averaging", C.c_uint8),
                ("SSB_HPF_pass", C.c_uint16), ("AutoGain", C.c_bool), ("ATT", C.c_bool), ("LPF", C.c_bool),
                ("BPF", C.c_bool), ("CW_Filter", C.c_uint16), ("SSB_Filter", C.c_uint16), ("FM_Filter", C.c_uint16)]


class Legacy:
    def __init__(self, L):
        self.L = L
        L.getPhraseFromFrequency.argtypes = [C.c_uint32]
        L.getPhraseFromFrequency.restype = C.c_uint32
        L.ddc_legacy_adc_push.argtypes = [C.c_void_p, C.c_size_t]
        L.ddc_legacy_adc_push.restype = C.c_size_t
        L.ddc_legacy_dac_pop.argtypes = [C.c_void_p, C.c_size_t]
        L.ddc_legacy_dac_pop.restype = C.c_size_t
        L.ddc_legacy_last_error.restype = C.c_char_p
        L.dc_filter.argtypes = [C.c_void_p, C.c_int16, C.c_uint8]
        L.DoAGC.argtypes = [C.c_void_p, C.c_int16]
        L.processNoiseReduction.argtypes = [C.c_void_p, C.c_void_p]
        L.CWDecoder_Process.argtypes = [C.c_void_p]
        L.TRX_setMode.argtypes = [C.c_uint8]
        for f in ("FPGA_Init", "FPGA_fpgadata_iqclock", "FPGA_fpgadata_stuffclock", "processRxAudio", "processTxAudio",
                  "initAudioProcessor", "FFT_Init", "FFT_doFFT", "ddc_legacy_shutdown", "dc_filter", "DoAGC",
                  "processNoiseReduction", "CWDecoder_Process", "TRX_DoAutoGain", "TRX_setMode"):
            getattr(L, f).restype = None
        self.TRX = TrxSettings.in_dll(L, "TRX")

    def arr(self, name, ctype, n):
        return np.ctypeslib.as_array((ctype * n).in_dll(self.L, name))

    def var(self, name, ctype):
        return ctype.in_dll(self.L, name)

    def err(self):
        return self.L.ddc_legacy_last_error().decode()
