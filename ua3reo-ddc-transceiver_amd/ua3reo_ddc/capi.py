"""ctypes binding of libua3reo_ddc.so.

The library is the product; this module only marshals pointers.  It fails
loudly when the shared object is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

DECIMATION = 1024
_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("UA3REO_DDC_LIB") or os.path.join(_PKG, "libua3reo_ddc.so")
_CSRC = os.path.join(_PKG, "csrc")


class DdcError(RuntimeError):
    pass


def lib_path():
    return _SO


def build_native(force=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))]
    inc = os.path.join(os.path.dirname(_PKG), "include")
    hdrs = [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    newest = max(os.path.getmtime(p) for p in srcs + hdrs)
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= newest:
        return _SO
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", _SO, os.path.join(_CSRC, "ddc_lib.hip"), os.path.join(_CSRC, "ddc_legacy.cpp")]
    subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise DdcError("native library %s is missing: run __graft_entry__.build() "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % _SO)
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7.  When
    # torch is installed, load it first so that this library binds to the same
    # runtime instance instead of bringing in a second one from /opt/rocm (two
    # runtimes in one process cannot both open the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if os.environ.get("UA3REO_DDC_LIB"):
        import sys
        sys.stderr.write("ua3reo_ddc: loading %s (UA3REO_DDC_LIB is set)\n" % _SO)
    L = C.CDLL(_SO)
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32
    L.ddc_create.argtypes = [vp, C.POINTER(vp)]
    L.ddc_destroy.argtypes = [vp]
    L.ddc_destroy.restype = None
    L.ddc_last_error.argtypes = [vp]
    L.ddc_last_error.restype = C.c_char_p
    L.ddc_version.restype = C.c_char_p
    L.ddc_tune.argtypes = [vp, i32, u32]
    L.ddc_tune_hz.argtypes = [vp, i32, u32, C.POINTER(C.c_int)]
    L.ddc_tune_many.argtypes = [vp, vp, u32]
    L.ddc_phrase_from_frequency.argtypes = [u32, u32, C.POINTER(C.c_int)]
    L.ddc_phrase_from_frequency.restype = u32
    L.ddc_frequency_from_phrase.argtypes = [u32, u32]
    L.ddc_frequency_from_phrase.restype = u32
    L.ddc_rx_reset.argtypes = [vp, i32]
    L.ddc_rx_set_phase.argtypes = [vp, i32, u32]
    L.ddc_rx_get_phase.argtypes = [vp, i32, C.POINTER(u32)]
    L.ddc_rx_process_dev.argtypes = [vp, vp, sz, sz, vp, vp]
    L.ddc_rx_process_dev_static.argtypes = [vp, vp, sz, sz, vp, vp]
    L.ddc_rx_static_scratch_bytes.argtypes = []
    L.ddc_rx_static_scratch_bytes.restype = sz
    L.ddc_rx_static_scratch_need.argtypes = [u32, sz]
    L.ddc_rx_static_scratch_need.restype = sz
    L.ddc_rx_process.argtypes = [vp, vp, sz, sz, vp, vp]
    L.ddc_rx_read_cic.argtypes = [vp, vp, sz]
    L.ddc_debug_rx_plan.argtypes = [vp, C.c_int]
    L.ddc_shard_range.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    L.ddc_shard_range.restype = None
    L.ddc_comm_unique_id.argtypes = [vp]
    L.ddc_comm_init.argtypes = [vp, u32, u32, vp]
    L.ddc_comm_attach.argtypes = [vp, vp, u32, u32]
    L.ddc_comm_destroy.argtypes = [vp]
    L.ddc_allgather_decimated_dev.argtypes = [vp, C.c_int, vp, sz, vp]
    L.ddc_allgather_reuse.argtypes = [vp, C.c_int]
    L.ddc_allgather_wait.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.ddc_tx_process_dev.argtypes = [vp, vp, sz, vp]
    L.ddc_tx_process.argtypes = [vp, vp, sz, vp]
    L.ddc_tx_reset.argtypes = [vp, i32]
    L.ddc_tx_set_phase.argtypes = [vp, i32, u32]
    L.ddc_sync.argtypes = [vp]
    L.ddc_stream.argtypes = [vp]
    L.ddc_stream.restype = vp
    L.ddc_set_stream.argtypes = [vp, vp]
    L.ddc_profile_enable.argtypes = [vp, C.c_int]
    L.ddc_profile_collect.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(u32)]
    L.ddc_pack_rx_frames.argtypes = [vp, vp, sz, vp]
    L.ddc_pack_rx_frames.restype = None
    L.ddc_unpack_tx_frames.argtypes = [vp, sz, vp]
    L.ddc_unpack_tx_frames.restype = None
    L.ddc_parse_params.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(u32)]
    L.ddc_parse_params.restype = None
    L.ddc_set_params.argtypes = [vp, i32, vp]
    L.ddc_pack_adc12.argtypes = [vp, sz, vp]
    L.ddc_debug_table.argtypes = [C.c_char_p, vp, sz, C.POINTER(sz)]
    L.ddc_debug_nco.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.ddc_synth_adc_dev.argtypes = [u32, u32, u32, sz, vp, sz, vp]
    L.ddc_pack_adc12.restype = None
    L.ddc_unpack_adc12.argtypes = [vp, sz, vp]
    L.ddc_unpack_adc12.restype = None
    L.ddc_adc_stats_enable.argtypes = [vp, C.c_int]
    L.ddc_get_status.argtypes = [vp, i32, vp]
    L.ddc_parse_status.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ddc_parse_status.restype = None
    L.ddc_state_size.argtypes = [vp]
    L.ddc_state_size.restype = sz
    L.ddc_state_save.argtypes = [vp, vp, sz]
    L.ddc_state_restore.argtypes = [vp, vp, sz]
    L.ddc_audio_configure.argtypes = [vp, i32, vp]
    L.ddc_audio_update.argtypes = [vp, i32, vp]
    L.ddc_audio_smeter_dbm.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.ddc_fft_get_waterfall.argtypes = [vp, i32, vp]
    L.ddc_autogain_step.argtypes = [vp, C.c_int16]
    L.ddc_fft_move_waterfall.argtypes = [vp, i32, i32]
    L.ddc_audio_stage.argtypes = [vp, i32, C.c_int, C.c_int, vp, C.c_int]
    L.ddc_audio_rx_process_dev.argtypes = [vp, vp, vp, sz, vp]
    L.ddc_audio_usb_frames_dev.argtypes = [vp, vp, sz, vp]
    L.ddc_audio_usb_frames.argtypes = [vp, vp, sz, vp]
    L.ddc_audio_params_for_mode.argtypes = [C.c_uint8, C.c_uint16, C.c_uint16, C.c_uint16, vp]
    L.ddc_audio_rx_process.argtypes = [vp, vp, vp, sz, vp]
    L.ddc_fft_process_dev.argtypes = [vp, vp, sz, vp, vp]
    L.ddc_fft_process.argtypes = [vp, vp, sz, vp, vp]
    L.ddc_audio_get_status.argtypes = [vp, i32, vp]
    L.ddc_audio_tx_process_dev.argtypes = [vp, vp, sz, vp, vp]
    L.ddc_audio_tx_process.argtypes = [vp, vp, sz, vp, vp]
    L.ddc_synth_adc.argtypes = [u32, u32, sz, vp]
    L.ddc_synth_adc.restype = None
    _lib = L
    return L


class _Config(C.Structure):
    _fields_ = [("n_channels", C.c_uint32), ("fs_adc", C.c_uint32), ("device", C.c_int32),
                ("reserved", C.c_uint32)]


MODES = {"LSB": 0, "USB": 1, "IQ": 2, "CW_L": 3, "CW_U": 4, "DIGI_L": 5, "DIGI_U": 6, "NO_TX": 7,
         "NFM": 8, "WFM": 9, "AM": 10, "LOOPBACK": 11}


class AudioParams(C.Structure):
    """ddc_audio_params: the TRX / VFO settings the audio path reads
    (STM32/Src/settings.h:38-101)."""
    _fields_ = [("mode", C.c_uint8), ("agc", C.c_uint8), ("dnr", C.c_uint8), ("notch", C.c_uint8),
                ("mute", C.c_uint8), ("cw_decoder", C.c_uint8), ("fm_sql_threshold", C.c_uint8),
                ("volume", C.c_uint8), ("iq_swap", C.c_uint8), ("fft_averaging", C.c_uint8),
                ("fft_zoom", C.c_uint8), ("rf_power", C.c_uint8), ("tune", C.c_uint8), ("key", C.c_uint8),
                ("pad0", C.c_uint8), ("pad1", C.c_uint8), ("filter_width", C.c_uint16),
                ("hpf_pass", C.c_uint16), ("notch_fc", C.c_uint16), ("agc_speed", C.c_uint16),
                ("rf_gain", C.c_float)]


class AudioStatus(C.Structure):
    _fields_ = [("smeter_max", C.c_float), ("smeter_min", C.c_float), ("agc_gain", C.c_float),
                ("fft_max_value", C.c_float), ("cw_wpm", C.c_uint32), ("squelched", C.c_uint32),
                ("blocks", C.c_uint32), ("alc_gain", C.c_float), ("tx_max_amplitude", C.c_float),
                ("cw_text", C.c_char * 16)]


def audio_params(mode="USB", agc=1, dnr=0, notch=0, mute=0, cw_decoder=0, fm_sql_threshold=0, volume=50,
                 iq_swap=0, fft_averaging=4, filter_width=2700, hpf_pass=300, notch_fc=1000, agc_speed=3,
                 rf_gain=50.0, fft_zoom=1, rf_power=50, tune=0, key=0):
    """Defaults follow LoadSettings (STM32/Src/settings.c:33-120) where it names one."""
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    return AudioParams(m, agc, dnr, notch, mute, cw_decoder, fm_sql_threshold, volume, iq_swap, fft_averaging,
                       fft_zoom, rf_power, tune, key, 0, 0, filter_width, hpf_pass, notch_fc, agc_speed, rf_gain)


def synth_adc_dev(first_channel, n_channels, samples, rows_ptr, row_bytes, fs_adc=0, stream=None):
    """Packed synthetic ADC rows generated on the GPU into device memory at rows_ptr."""
    L = lib()
    rc = L.ddc_synth_adc_dev(first_channel, n_channels, fs_adc, samples, rows_ptr, row_bytes, stream)
    if rc != 0:
        raise DdcError("error %d: %s" % (rc, L.ddc_last_error(None).decode()))


def debug_table(name, dtype):
    """A constant table as compiled into the library (ddc_debug_table), as a flat numpy array."""
    L = lib()
    n = C.c_size_t(0)
    rc = L.ddc_debug_table(name.encode(), None, 0, C.byref(n))
    if rc != 0:
        raise DdcError("error %d: %s" % (rc, L.ddc_last_error(None).decode()))
    out = np.zeros(n.value // np.dtype(dtype).itemsize, dtype)
    rc = L.ddc_debug_table(name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(n))
    if rc != 0:
        raise DdcError("error %d: %s" % (rc, L.ddc_last_error(None).decode()))
    return out


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Ddc:
    """One ddc_ctx: N tuned channels on one GPU."""

    def __init__(self, n_channels, fs_adc=0, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        cfg = _Config(n_channels, fs_adc, device, 0)
        rc = self._L.ddc_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise DdcError("ddc_create failed (%d): %s" % (rc, self._L.ddc_last_error(None).decode()))
        self.n_channels = n_channels

    def close(self):
        if self._h:
            self._L.ddc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DdcError("error %d: %s" % (rc, self._L.ddc_last_error(self._h).decode()))

    @property
    def handle(self):
        return self._h

    def tune(self, channel, phi):
        self._check(self._L.ddc_tune(self._h, channel, int(phi)))

    def tune_many(self, phis):
        a = np.ascontiguousarray(phis, dtype=np.uint32)
        self._check(self._L.ddc_tune_many(self._h, _np_ptr(a), a.size))

    def tune_hz(self, channel, freq):
        sw = C.c_int()
        self._check(self._L.ddc_tune_hz(self._h, channel, int(freq), C.byref(sw)))
        return bool(sw.value)

    def rx_reset(self, channel=-1):
        self._check(self._L.ddc_rx_reset(self._h, channel))

    def rx_set_phase(self, channel, phase):
        self._check(self._L.ddc_rx_set_phase(self._h, channel, int(phase)))

    def rx_get_phase(self, channel):
        v = C.c_uint32()
        self._check(self._L.ddc_rx_get_phase(self._h, channel, C.byref(v)))
        return v.value

    def rx_process(self, packed_rows, samples, row_bytes=None):
        """Host path. packed_rows: uint8 [n_channels][row_bytes]; returns spec, voice
        int16 [n_channels][samples/1024][2].  row_bytes=0: packed_rows is one row [1][bytes]
        shared by every channel."""
        a = np.ascontiguousarray(packed_rows, dtype=np.uint8)
        if row_bytes is None:
            row_bytes = a.shape[1]
        assert a.ndim == 2 and a.shape[0] == (self.n_channels if row_bytes else 1)
        nf = samples // DECIMATION
        spec = np.zeros((self.n_channels, nf, 2), np.int16)
        voice = np.zeros((self.n_channels, nf, 2), np.int16)
        self._check(self._L.ddc_rx_process(self._h, _np_ptr(a), row_bytes, samples, _np_ptr(spec),
                                           _np_ptr(voice)))
        return spec, voice

    def rx_process_dev(self, adc_ptr, row_bytes, samples, spec_ptr, voice_ptr):
        """Device path: raw device addresses (ints), asynchronous on the ctx stream."""
        self._check(self._L.ddc_rx_process_dev(self._h, adc_ptr, row_bytes, samples, spec_ptr, voice_ptr))

    def rx_process_dev_static(self, adc_ptr, row_bytes, samples, spec_ptr, voice_ptr):
        """ddc_rx_process_dev with the call's work buffers in the library's fixed
        per-GPU work area: allocates nothing; DdcError (-E2BIG) beyond its size."""
        self._check(self._L.ddc_rx_process_dev_static(self._h, adc_ptr, row_bytes, samples, spec_ptr, voice_ptr))

    def debug_nco(self, first_phase, count):
        """(sin14, cos14, sin12, cos12) of the kernels' NCO for `count` consecutive 22-bit phases:
        the TX kernel's 14-bit values and the RX kernel's 12-bit mixer operands (int16 arrays)."""
        out = [np.zeros(count, np.int16) for _ in range(4)]
        self._check(self._L.ddc_debug_nco(self._h, first_phase, count, *[_np_ptr(a) for a in out]))
        return tuple(out)

    def rx_read_cic(self, samples):
        out = np.zeros((self.n_channels, samples // 512, 2), np.int16)
        self._check(self._L.ddc_rx_read_cic(self._h, _np_ptr(out), samples))
        return out

    def debug_rx_plan(self, plan):
        """0: call plan by size (one kernel when samples / 512 is a multiple of 256); 1: always the separate kernels."""
        self._check(self._L.ddc_debug_rx_plan(self._h, int(plan)))

    def comm_init(self, world, rank, comm_id):
        """Collective: builds this ctx's RCCL communicator from the 128-byte id of comm_unique_id() (the same on every rank)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(comm_id))
        self._check(self._L.ddc_comm_init(self._h, world, rank, buf))

    def comm_destroy(self):
        self._check(self._L.ddc_comm_destroy(self._h))

    def allgather_decimated_dev(self, slot, local_ptr, n_frames, gathered_ptr):
        self._check(self._L.ddc_allgather_decimated_dev(self._h, slot, local_ptr, n_frames, gathered_ptr))

    def allgather_reuse(self, slot):
        self._check(self._L.ddc_allgather_reuse(self._h, slot))

    def allgather_wait(self, slot=-1):
        """Host wait for slot's gather (-1: both); returns the duration in ms of the last gather waited for."""
        ms = C.c_float(0)
        self._check(self._L.ddc_allgather_wait(self._h, slot, C.byref(ms)))
        return ms.value

    def tx_process(self, iq48k):
        """Host path. iq48k: int16 [n_channels][n][2]; returns uint16 [n_channels][1024 n]."""
        a = np.ascontiguousarray(iq48k, dtype=np.int16)
        assert a.ndim == 3 and a.shape[0] == self.n_channels and a.shape[2] == 2
        n = a.shape[1]
        dac = np.zeros((self.n_channels, 1024 * n), np.uint16)
        self._check(self._L.ddc_tx_process(self._h, _np_ptr(a), n, _np_ptr(dac)))
        return dac

    def tx_process_dev(self, iq_ptr, n_iq, dac_ptr):
        self._check(self._L.ddc_tx_process_dev(self._h, iq_ptr, n_iq, dac_ptr))

    def tx_reset(self, channel=-1):
        self._check(self._L.ddc_tx_reset(self._h, channel))

    def tx_set_phase(self, channel, phase):
        self._check(self._L.ddc_tx_set_phase(self._h, channel, int(phase)))

    def audio_configure(self, params, channel=-1):
        self._check(self._L.ddc_audio_configure(self._h, channel, C.byref(params)))

    def audio_update(self, params, channel=-1):
        self._check(self._L.ddc_audio_update(self._h, channel, C.byref(params)))

    def audio_stage(self, channel, stage, arg, buf):
        """One firmware stage (0 dc_filter, 1 DoAGC, 2 noise reduction, 3 CW decoder) in place on float32 buf."""
        b = np.ascontiguousarray(buf, dtype=np.float32).copy()
        self._check(self._L.ddc_audio_stage(self._h, channel, stage, arg, _np_ptr(b), b.size))
        return b

    def audio_rx_process(self, spec_iq, voice_iq):
        """Host path. spec/voice int16 [n_channels][n][2] -> int32 [n_channels][n][2] (L, R)."""
        s = np.ascontiguousarray(spec_iq, dtype=np.int16)
        v = np.ascontiguousarray(voice_iq, dtype=np.int16)
        assert s.shape == v.shape and s.ndim == 3 and s.shape[0] == self.n_channels and s.shape[2] == 2
        out = np.zeros(s.shape, np.int32)
        self._check(self._L.ddc_audio_rx_process(self._h, _np_ptr(s), _np_ptr(v), s.shape[1], _np_ptr(out)))
        return out

    def audio_usb_frames(self, audio_lr):
        """int32 [n_channels][n_frames][2] (host) -> the int16 USB-audio frames of the same shape."""
        a = np.ascontiguousarray(audio_lr, dtype=np.int32)
        assert a.ndim == 3 and a.shape[0] == self.n_channels and a.shape[2] == 2
        out = np.zeros(a.shape, np.int16)
        self._check(self._L.ddc_audio_usb_frames(self._h, _np_ptr(a), a.shape[1], _np_ptr(out)))
        return out

    def audio_usb_frames_dev(self, audio_ptr, n_frames, usb_ptr):
        self._check(self._L.ddc_audio_usb_frames_dev(self._h, audio_ptr, n_frames, usb_ptr))

    def audio_rx_process_dev(self, spec_ptr, voice_ptr, n_frames, audio_ptr):
        self._check(self._L.ddc_audio_rx_process_dev(self._h, spec_ptr, voice_ptr, n_frames, audio_ptr))

    def audio_tx_process(self, mic_lr):
        """Host path. mic int16 [n_channels][n][2] (L, R) -> (iq int16 [n_channels][n][2],
        loopback int32 [n_channels][n][2])."""
        m = np.ascontiguousarray(mic_lr, dtype=np.int16)
        assert m.ndim == 3 and m.shape[0] == self.n_channels and m.shape[2] == 2
        iq = np.zeros(m.shape, np.int16)
        loop = np.zeros(m.shape, np.int32)
        self._check(self._L.ddc_audio_tx_process(self._h, _np_ptr(m), m.shape[1], _np_ptr(iq), _np_ptr(loop)))
        return iq, loop

    def audio_tx_process_dev(self, mic_ptr, n_frames, iq_ptr, loop_ptr=None):
        self._check(self._L.ddc_audio_tx_process_dev(self._h, mic_ptr, n_frames, iq_ptr, loop_ptr))

    def fft_process(self, spec_iq):
        """Host path. spec int16 [n_channels][n][2], n % 512 == 0 -> (bins float32
        [n_channels][n/512][256], rows uint16 [n_channels][n/512][256])."""
        s = np.ascontiguousarray(spec_iq, dtype=np.int16)
        assert s.ndim == 3 and s.shape[0] == self.n_channels and s.shape[2] == 2
        nf = s.shape[1] // 512
        bins = np.zeros((self.n_channels, nf, 256), np.float32)
        rows = np.zeros((self.n_channels, nf, 256), np.uint16)
        self._check(self._L.ddc_fft_process(self._h, _np_ptr(s), s.shape[1], _np_ptr(bins), _np_ptr(rows)))
        return bins, rows

    def smeter_dbm(self, channel):
        v = C.c_float()
        self._check(self._L.ddc_audio_smeter_dbm(self._h, channel, C.byref(v)))
        return v.value

    def fft_waterfall(self, channel):
        out = np.zeros((50, 256), np.uint16)
        self._check(self._L.ddc_fft_get_waterfall(self._h, channel, _np_ptr(out)))
        return out

    def fft_move_waterfall(self, channel, freq_diff_hz):
        self._check(self._L.ddc_fft_move_waterfall(self._h, channel, int(freq_diff_hz)))

    def fft_process_dev(self, spec_ptr, n_frames, bins_ptr, rows_ptr):
        self._check(self._L.ddc_fft_process_dev(self._h, spec_ptr, n_frames, bins_ptr, rows_ptr))

    def audio_status(self, channel):
        st = AudioStatus()
        self._check(self._L.ddc_audio_get_status(self._h, channel, C.byref(st)))
        return {"smeter_max": st.smeter_max, "smeter_min": st.smeter_min, "agc_gain": st.agc_gain,
                "fft_max_value": st.fft_max_value, "cw_wpm": st.cw_wpm, "squelched": bool(st.squelched),
                "blocks": st.blocks, "alc_gain": st.alc_gain, "tx_max_amplitude": st.tx_max_amplitude, "cw_text": bytes(st.cw_text).split(b"\0")[0].decode("ascii", "replace")}

    def adc_stats_enable(self, on=True):
        self._check(self._L.ddc_adc_stats_enable(self._h, 1 if on else 0))

    def get_status(self, channel):
        """The 5-byte command-2 reply of the bus for one channel (resets ADC min/max)."""
        out = np.zeros(5, np.uint8)
        self._check(self._L.ddc_get_status(self._h, channel, _np_ptr(out)))
        return out

    def set_params(self, frame, channel=-1):
        """Command-1 payload (4 bytes: flags, tuning word) for one channel or all."""
        b = np.ascontiguousarray(frame, dtype=np.uint8)
        assert b.size == 4
        self._check(self._L.ddc_set_params(self._h, channel, _np_ptr(b)))

    def state_save(self):
        n = self._L.ddc_state_size(self._h)
        buf = np.zeros(n, np.uint8)
        self._check(self._L.ddc_state_save(self._h, _np_ptr(buf), n))
        return buf

    def state_restore(self, blob):
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self._L.ddc_state_restore(self._h, _np_ptr(b), b.size))

    def set_stream(self, hip_stream):
        """Enqueue on the caller's hipStream_t (an int handle, e.g. torch.cuda.current_stream().cuda_stream); None = own stream."""
        self._check(self._L.ddc_set_stream(self._h, hip_stream))

    def sync(self):
        self._check(self._L.ddc_sync(self._h))

    def profile_enable(self, on=True):
        self._check(self._L.ddc_profile_enable(self._h, 1 if on else 0))

    def profile_collect(self):
        """(front kernel ms summed, rest-of-call ms summed, calls)"""
        f, t, n = C.c_double(), C.c_double(), C.c_uint32()
        self._check(self._L.ddc_profile_collect(self._h, C.byref(f), C.byref(t), C.byref(n)))
        return f.value, t.value, n.value

    def stream(self):
        return self._L.ddc_stream(self._h)


def shard_range(rank, world, total_channels):
    """[lo, hi) of the channels `rank` owns (ddc_shard_range)."""
    lo, hi = C.c_uint32(0), C.c_uint32(0)
    lib().ddc_shard_range(rank, world, total_channels, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def comm_unique_id():
    """128 bytes that name a new RCCL communicator: taken on one rank, handed to every rank's Ddc.comm_init."""
    buf = (C.c_uint8 * 128)()
    rc = lib().ddc_comm_unique_id(buf)
    if rc != 0:
        raise DdcError("ddc_comm_unique_id failed (%d): %s" % (rc, (lib().ddc_last_error(None) or b"").decode()))
    return bytes(buf)


def pack_adc12(samples):
    s = np.ascontiguousarray(samples, dtype=np.int16)
    out = np.zeros((s.size + 1) // 2 * 3, np.uint8)
    lib().ddc_pack_adc12(_np_ptr(s), s.size, _np_ptr(out))
    return out


def unpack_adc12(packed, n):
    p = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.zeros(n, np.int16)
    lib().ddc_unpack_adc12(_np_ptr(p), n, _np_ptr(out))
    return out


def synth_adc(channel, n, fs_adc=0):
    out = np.zeros(n, np.int16)
    lib().ddc_synth_adc(channel, fs_adc, n, _np_ptr(out))
    return out


def phrase_from_frequency(freq, fs_adc=0):
    sw = C.c_int()
    v = lib().ddc_phrase_from_frequency(int(freq), fs_adc, C.byref(sw))
    return int(v), bool(sw.value)


def frequency_from_phrase(phrase, fs_adc=0):
    return int(lib().ddc_frequency_from_phrase(int(phrase), fs_adc))


def pack_rx_frames(spec_iq, voice_iq):
    s = np.ascontiguousarray(spec_iq, dtype=np.int16).reshape(-1, 2)
    v = np.ascontiguousarray(voice_iq, dtype=np.int16).reshape(-1, 2)
    out = np.zeros(8 * s.shape[0], np.uint8)
    lib().ddc_pack_rx_frames(_np_ptr(s), _np_ptr(v), s.shape[0], _np_ptr(out))
    return out


def unpack_tx_frames(frames):
    f = np.ascontiguousarray(frames, dtype=np.uint8)
    n = f.size // 4
    out = np.zeros((n, 2), np.int16)
    lib().ddc_unpack_tx_frames(_np_ptr(f), n, _np_ptr(out))
    return out


def parse_status(b):
    """(adc_otr, dac_otr, adc_min, adc_max) of a 5-byte status frame."""
    buf = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    a, d, lo, hi = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().ddc_parse_status(_np_ptr(buf), C.byref(a), C.byref(d), C.byref(lo), C.byref(hi))
    return a.value, d.value, lo.value, hi.value


def parse_params(b):
    buf = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    tx, pre, phi = C.c_int(), C.c_int(), C.c_uint32()
    lib().ddc_parse_params(_np_ptr(buf), C.byref(tx), C.byref(pre), C.byref(phi))
    return tx.value, pre.value, phi.value


class AutoGain:
    """ddc_autogain_step: the firmware's once-a-second preamp / attenuator search."""

    def __init__(self):
        self._st = (C.c_uint8 * 4)()

    def step(self, adc_max_amplitude):
        """Returns True when preamp / att changed."""
        rc = lib().ddc_autogain_step(self._st, int(adc_max_amplitude))
        if rc < 0:
            raise DdcError("ddc_autogain_step failed: %d" % rc)
        return bool(rc)

    @property
    def state(self):
        return {"stage": self._st[0], "wait_reaction": self._st[1], "preamp": bool(self._st[2]), "att": bool(self._st[3])}
