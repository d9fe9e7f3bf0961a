"""ctypes front end of the CPU oracle (oracle/ddc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py, never by the shipped package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libddc_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ddc_oracle.c", "audio_oracle.c", "oracle_tables.h", "oracle_cic_consts.h",
                                          "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(p) for p in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.orc_init.restype = None
        for name in ("orc_rx_front_sizeof", "orc_rx_ciccomp_sizeof", "orc_rx_hilb_sizeof",
                     "orc_delay_sizeof", "orc_rx_chain_sizeof", "orc_tx_chain_sizeof"):
            if hasattr(L, name):
                getattr(L, name).restype = C.c_size_t
        L.orc_rx_front_run.restype = C.c_size_t
        L.orc_rx_ciccomp_run.restype = C.c_size_t
        L.orc_rx_ciccomp_cycle.restype = C.c_size_t
        L.orc_rx_hilb_run.restype = C.c_size_t
        L.orc_rx_hilb_cycle.restype = C.c_size_t
        L.orc_delay_run.restype = C.c_size_t
        L.orc_rx_chain_run.restype = C.c_size_t
        L.orc_phrase_from_freq.restype = C.c_uint32
        L.orc_freq_from_phrase.restype = C.c_uint32
        L.orc_arm_sin_f32.restype = C.c_float
        L.orc_arm_sin_f32.argtypes = [C.c_float]
        L.orc_arm_cos_f32.restype = C.c_float
        L.orc_arm_cos_f32.argtypes = [C.c_float]
        L.orc_arm_sin_table.restype = C.POINTER(C.c_float)
        L.orc_init()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def nco_tables():
    t = [np.zeros(2048, np.int16) for _ in range(4)]
    lib().orc_nco_tables(*[_p(x) for x in t])
    return t  # sc, cc, sf, cf


def nco_eval(acc):
    s, c = C.c_int(), C.c_int()
    lib().orc_nco_eval(C.c_uint32(int(acc)), C.byref(s), C.byref(c))
    return s.value, c.value


class _State:
    _size = None
    _reset = None

    def __init__(self, *args):
        L = lib()
        self.buf = C.create_string_buffer(getattr(L, self._size)())
        getattr(L, self._reset)(self.buf, *args)


class RxFront(_State):
    """mixer.v + rx_mixer_shift.v + rx_cic.vhd, clock-edge model, I and Q."""
    _size, _reset = "orc_rx_front_sizeof", "orc_rx_front_reset"

    def __init__(self, phi):
        super().__init__(C.c_uint32(int(phi)))

    def tune(self, phi):
        lib().orc_rx_front_tune(self.buf, C.c_uint32(int(phi)))

    def run(self, adc):
        adc = _i16(adc)
        n = adc.size
        oi = np.zeros(n // 512 + 2, np.int16)
        oq = np.zeros(n // 512 + 2, np.int16)
        k = lib().orc_rx_front_run(self.buf, _p(adc), C.c_size_t(n), _p(oi), _p(oq))
        return oi[:k].copy(), oq[:k].copy()


class RxCicComp(_State):
    _size, _reset = "orc_rx_ciccomp_sizeof", "orc_rx_ciccomp_reset"

    def run(self, x):
        x = _i16(x)
        out = np.zeros(x.size // 2 + 2, np.int16)
        k = lib().orc_rx_ciccomp_run(self.buf, _p(x), C.c_size_t(x.size), _p(out))
        return out[:k].copy()


def rx_ciccomp_cycle(x):
    x = _i16(x)
    out = np.zeros(x.size // 2 + 2, np.int16)
    k = lib().orc_rx_ciccomp_cycle(_p(x), C.c_size_t(x.size), _p(out), C.c_size_t(out.size))
    return out[:min(k, out.size)].copy()


class RxHilb(_State):
    _size, _reset = "orc_rx_hilb_sizeof", "orc_rx_hilb_reset"

    def run(self, x):
        x = _i16(x)
        out = np.zeros(x.size, np.int16)
        lib().orc_rx_hilb_run(self.buf, _p(x), C.c_size_t(x.size), _p(out))
        return out


def rx_hilb_cycle(x):
    x = _i16(x)
    out = np.zeros(x.size + 2, np.int16)
    k = lib().orc_rx_hilb_cycle(_p(x), C.c_size_t(x.size), _p(out), C.c_size_t(out.size))
    return out[:min(k, out.size)].copy()


class Delay130(_State):
    _size, _reset = "orc_delay_sizeof", "orc_delay_reset"

    def run(self, x):
        x = _i16(x)
        out = np.zeros(x.size, np.int16)
        lib().orc_delay_run(self.buf, _p(x), C.c_size_t(x.size), _p(out))
        return out


class RxChain(_State):
    """Whole FPGA RX side for one channel: ADC words -> SPEC/VOICE I/Q."""
    _size, _reset = "orc_rx_chain_sizeof", "orc_rx_chain_reset"

    def __init__(self, phi):
        super().__init__(C.c_uint32(int(phi)))

    def tune(self, phi):
        lib().orc_rx_chain_tune(self.buf, C.c_uint32(int(phi)))

    def run(self, adc, want_cic=False):
        adc = _i16(adc)
        n = adc.size
        assert n % 1024 == 0
        nf = n // 1024
        spec = np.zeros((nf, 2), np.int16)
        voice = np.zeros((nf, 2), np.int16)
        cic = np.zeros((2 * nf, 2), np.int16) if want_cic else None
        k = lib().orc_rx_chain_run(self.buf, _p(adc), C.c_size_t(n), _p(spec), _p(voice),
                                   _p(cic) if want_cic else None)
        assert k == nf
        return (spec, voice, cic) if want_cic else (spec, voice)


def build_native_variant(out_dir):
    """The same sources built for the host this runs on (gcc -O3 -march=native, widest vectors), for
    timing the CPU baseline; returns a CDLL of its own, ready to use."""
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libddc_oracle_native.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-mprefer-vector-width=512", "-fopenmp", "-fPIC", "-std=c11",
                           "-ffp-contract=off", "-shared", "-o", so,
                           os.path.join(_HERE, "ddc_oracle.c"), os.path.join(_HERE, "audio_oracle.c"), "-lm"])
    L = C.CDLL(so)
    L.orc_init.restype = None
    L.orc_init()
    return L


def rx_chain_packed(packed, row_bytes, n_channels, n, phi, L=None, fast=False):
    """Fresh-state RX chain over packed 12-bit rows; returns spec, voice [ch][frame][2].
    fast: the window form of mixer + rx_cic (orc_fast_rx_front_run) instead of the clock-level model."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    phi = np.ascontiguousarray(phi, dtype=np.uint32)
    nf = n // 1024
    spec = np.zeros((n_channels, nf, 2), np.int16)
    voice = np.zeros((n_channels, nf, 2), np.int16)
    f = (L or lib()).orc_fast_rx_chain_packed if fast else (L or lib()).orc_rx_chain_packed
    f(_p(packed), C.c_size_t(row_bytes), C.c_size_t(n_channels), C.c_size_t(n), _p(phi), _p(spec), _p(voice))
    return spec, voice


def rx_chain_packed_mt(L, packed, row_bytes, n_rows, n_channels, n, phi, threads=0):
    """orc_fast_rx_chain_packed_mt of a build_native_variant() library: the window-form RX chain
    over OpenMP threads, channel ch reading row ch % n_rows.  Returns spec, voice, threads used."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    phi = np.ascontiguousarray(phi, dtype=np.uint32)
    nf = n // 1024
    spec = np.zeros((n_channels, nf, 2), np.int16)
    voice = np.zeros((n_channels, nf, 2), np.int16)
    L.orc_fast_rx_chain_packed_mt.restype = C.c_int
    used = L.orc_fast_rx_chain_packed_mt(_p(packed), C.c_size_t(row_bytes), C.c_size_t(n_rows), C.c_size_t(n_channels),
                                         C.c_size_t(n), _p(phi), _p(spec), _p(voice), C.c_int(threads))
    return spec, voice, used


def pack12(x):
    x = _i16(x)
    out = np.zeros((x.size + 1) // 2 * 3, np.uint8)
    lib().orc_pack12(_p(x), C.c_size_t(x.size), _p(out))
    return out


def unpack12(packed, n):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.zeros(n, np.int16)
    lib().orc_unpack12(_p(packed), C.c_size_t(n), _p(out))
    return out


def phrase_from_freq(freq, adc_clock=49152000):
    sw = C.c_int()
    v = lib().orc_phrase_from_freq(C.c_uint32(int(freq)), C.c_uint32(int(adc_clock)), C.byref(sw))
    return int(v), bool(sw.value)


def freq_from_phrase(phrase, adc_clock=49152000):
    return int(lib().orc_freq_from_phrase(C.c_uint32(int(phrase)), C.c_uint32(int(adc_clock))))


def pack_rx_frame(spec_i, spec_q, voice_i, voice_q):
    out = (C.c_uint8 * 8)()
    lib().orc_pack_rx_frame(C.c_int16(spec_i), C.c_int16(spec_q), C.c_int16(voice_i),
                            C.c_int16(voice_q), out)
    return bytes(out)


def unpack_tx_frame(b):
    buf = (C.c_uint8 * 4)(*b)
    i, q = C.c_int16(), C.c_int16()
    lib().orc_unpack_tx_frame(buf, C.byref(i), C.byref(q))
    return i.value, q.value


def parse_params(b):
    buf = (C.c_uint8 * 4)(*b)
    tx, pre, f = C.c_int(), C.c_int(), C.c_uint32()
    lib().orc_parse_params(buf, C.byref(tx), C.byref(pre), C.byref(f))
    return tx.value, pre.value, f.value


# ------------------------------------------------------------------ TX ----
def _tx_proto():
    L = lib()
    L.orc_tx_ciccomp_sizeof.restype = C.c_size_t
    L.orc_tx_back_sizeof.restype = C.c_size_t
    L.orc_tx_chain_sizeof.restype = C.c_size_t
    L.orc_tx_ciccomp_run.restype = C.c_size_t
    L.orc_tx_ciccomp_cycle.restype = C.c_size_t
    L.orc_tx_back_run.restype = C.c_size_t
    L.orc_tx_chain_run.restype = C.c_size_t
    return L


class TxCicComp(_State):
    _size, _reset = "orc_tx_ciccomp_sizeof", "orc_tx_ciccomp_reset"

    def __init__(self):
        _tx_proto()
        super().__init__()

    def run(self, x):
        x = _i16(x)
        out = np.zeros(2 * x.size, np.int16)
        lib().orc_tx_ciccomp_run(self.buf, _p(x), C.c_size_t(x.size), _p(out))
        return out


def tx_ciccomp_cycle(x):
    _tx_proto()
    x = _i16(x)
    out = np.zeros(2 * x.size + 4, np.int16)
    k = lib().orc_tx_ciccomp_cycle(_p(x), C.c_size_t(x.size), _p(out), C.c_size_t(out.size))
    return out[:min(k, out.size)].copy()


class TxBack(_State):
    """tx_cic + tx_mixer + tx_summator + DAC_corrector, clock-edge model."""
    _size, _reset = "orc_tx_back_sizeof", "orc_tx_back_reset"

    def __init__(self, phi):
        _tx_proto()
        super().__init__(C.c_uint32(int(phi)))

    def set_phase(self, acc):
        lib().orc_tx_back_set_phase(self.buf, C.c_uint32(int(acc)))

    def run(self, yi, yq, want_cic=False):
        yi, yq = _i16(yi), _i16(yq)
        n = yi.size
        dac = np.zeros(512 * n, np.uint16)
        ci = np.zeros(512 * n, np.int16) if want_cic else None
        cq = np.zeros(512 * n, np.int16) if want_cic else None
        lib().orc_tx_back_run(self.buf, _p(yi), _p(yq), C.c_size_t(n), _p(dac),
                              _p(ci) if want_cic else None, _p(cq) if want_cic else None)
        return (dac, ci, cq) if want_cic else dac

    def overflow(self):
        return bool(lib().orc_tx_back_overflow(self.buf))


class TxChain(_State):
    """Whole FPGA TX side for one channel: 48 kHz I/Q -> DAC words."""
    _size, _reset = "orc_tx_chain_sizeof", "orc_tx_chain_reset"

    def __init__(self, phi):
        _tx_proto()
        super().__init__(C.c_uint32(int(phi)))

    def tune(self, phi):
        lib().orc_tx_chain_tune(self.buf, C.c_uint32(int(phi)))

    def set_phase(self, acc):
        lib().orc_tx_chain_set_phase(self.buf, C.c_uint32(int(acc)))

    def run(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        n = iq.shape[0]
        dac = np.zeros(1024 * n, np.uint16)
        lib().orc_tx_chain_run(self.buf, _p(iq), C.c_size_t(n), _p(dac))
        return dac

    def overflow(self):
        return bool(lib().orc_tx_chain_overflow(self.buf))


# --------------------------------------------------------------- audio ----
MODES = {"LSB": 0, "USB": 1, "IQ": 2, "CW_L": 3, "CW_U": 4, "DIGI_L": 5, "DIGI_U": 6, "NO_TX": 7,
         "NFM": 8, "WFM": 9, "AM": 10, "LOOPBACK": 11}


class AudioParams(C.Structure):
    _fields_ = [("mode", C.c_uint8), ("agc", C.c_uint8), ("dnr", C.c_uint8), ("notch", C.c_uint8),
                ("mute", C.c_uint8), ("cw_decoder", C.c_uint8), ("fm_sql_threshold", C.c_uint8),
                ("volume", C.c_uint8), ("iq_swap", C.c_uint8), ("fft_averaging", C.c_uint8),
                ("fft_zoom", C.c_uint8), ("rf_power", C.c_uint8), ("tune", C.c_uint8), ("key", C.c_uint8),
                ("pad0", C.c_uint8), ("pad1", C.c_uint8), ("filter_width", C.c_uint16),
                ("hpf_pass", C.c_uint16), ("notch_fc", C.c_uint16), ("agc_speed", C.c_uint16),
                ("rf_gain", C.c_float)]


def audio_params(mode="USB", agc=1, dnr=0, notch=0, mute=0, cw_decoder=0, fm_sql_threshold=0, volume=50,
                 iq_swap=0, fft_averaging=4, filter_width=2700, hpf_pass=300, notch_fc=1000, agc_speed=3,
                 rf_gain=50.0, fft_zoom=1, rf_power=50, tune=0, key=0):
    return AudioParams(MODES[mode] if isinstance(mode, str) else mode, agc, dnr, notch, mute, cw_decoder,
                       fm_sql_threshold, volume, iq_swap, fft_averaging, fft_zoom, rf_power, tune, key, 0, 0,
                       filter_width, hpf_pass, notch_fc, agc_speed, rf_gain)


class Audio:
    """STM32 RX audio path + FFT panadapter of one channel."""

    def __init__(self, params):
        L = lib()
        L.orc_audio_sizeof.restype = C.c_size_t
        self.buf = C.create_string_buffer(L.orc_audio_sizeof())
        if L.orc_audio_configure(self.buf, C.byref(params)) != 0:
            raise ValueError("unsupported filter selection")

    def rx(self, spec_iq, voice_iq):
        spec = np.ascontiguousarray(spec_iq, dtype=np.int16).reshape(-1, 2)
        voice = np.ascontiguousarray(voice_iq, dtype=np.int16).reshape(-1, 2)
        n = spec.shape[0]
        assert n % 192 == 0 and voice.shape[0] == n
        out = np.zeros((n, 2), np.int32)
        lib().orc_audio_rx_run(self.buf, _p(spec), _p(voice), C.c_size_t(n), _p(out))
        return out

    def status(self):
        mx, mn, g = C.c_float(), C.c_float(), C.c_float()
        text = C.create_string_buffer(16)
        wpm, sq = C.c_uint16(), C.c_int()
        lib().orc_audio_status(self.buf, C.byref(mx), C.byref(mn), C.byref(g), text, C.byref(wpm), C.byref(sq))
        return {"smeter_max": mx.value, "smeter_min": mn.value, "agc_gain": g.value,
                "cw_text": text.value.decode("ascii", "replace"), "cw_wpm": wpm.value, "squelched": bool(sq.value)}

    def smeter_dbm(self):
        f = lib().orc_audio_smeter_dbm
        f.restype = C.c_float
        return float(f(self.buf))

    def waterfall(self):
        out = np.zeros((50, 256), np.uint16)
        lib().orc_fft_get_waterfall(self.buf, _p(out))
        return out

    def move_waterfall(self, freq_diff):
        lib().orc_fft_move_waterfall(self.buf, C.c_int16(int(freq_diff)))

    def tx(self, mic_lr):
        """processTxAudio over n frames (n % 192 == 0): returns (iq int16 [n][2], loopback int32 [n][2])."""
        mic = np.ascontiguousarray(mic_lr, dtype=np.int16).reshape(-1, 2)
        n = mic.shape[0]
        assert n % 192 == 0
        iq = np.zeros((n, 2), np.int16)
        loop = np.zeros((n, 2), np.int32)
        lib().orc_audio_tx_run(self.buf, _p(mic), C.c_size_t(n), _p(iq), _p(loop))
        return iq, loop

    def alc_gain(self):
        f = lib().orc_audio_alc_gain
        f.restype = C.c_float
        return float(f(self.buf))

    def fft(self, spec_iq):
        spec = np.ascontiguousarray(spec_iq, dtype=np.int16).reshape(-1, 2)
        assert spec.shape[0] == 512
        bins = np.zeros(256, np.float32)
        row = np.zeros(256, np.uint16)
        hts = np.zeros(256, np.uint8)
        lib().orc_fft_frame(self.buf, _p(spec), _p(bins), _p(row), _p(hts))
        return bins, row, hts


def usb_frames(audio_lr, volume):
    """audio_processor.c:415-432 on an int32 array of any shape -> int16 of the same shape."""
    a = np.ascontiguousarray(audio_lr, dtype=np.int32)
    out = np.zeros(a.shape, np.int16)
    lib().orc_audio_usb_frames(_p(a), C.c_size_t(a.size), C.c_uint8(volume), _p(out))
    return out


def params_for_mode(mode, ssb_filter, cw_filter, fm_filter, filter_width):
    """TRX_setMode (trx_manager.c:193-220): the Filter_Width the mode selects."""
    if mode in (0, 1, 5, 6, 10):
        return ssb_filter
    if mode in (3, 4):
        return cw_filter
    if mode == 8:
        return fm_filter
    if mode == 9:
        return 0
    return filter_width


def autogain_step(state, adc_max):
    """TRX_DoAutoGain (trx_manager.c:268-356) on state = [stage, wait, preamp, att] (list of 4 ints, updated in place)."""
    st = (C.c_uint8 * 4)(*state)
    lib().orc_autogain_step(st, C.c_int16(int(adc_max)))
    state[:] = list(st)
    return state


def fast_rx_front(phi, adc):
    """orc_fast_rx_front_run: window form of mixer + rx_cic from reset; returns (I, Q) int16 arrays of n // 512 outputs."""
    adc = _i16(adc)
    n = adc.size
    oi = np.zeros(n // 512, np.int16)
    oq = np.zeros(n // 512, np.int16)
    f = lib().orc_fast_rx_front_run
    f.restype = C.c_size_t
    f(C.c_uint32(int(phi)), _p(adc), C.c_size_t(n), _p(oi), _p(oq))
    return oi, oq


def arm_sin_f32(x):
    """arm_sin_f32 of CMSIS-DSP as the oracle restates it (513-entry table, linear interpolation)."""
    return float(lib().orc_arm_sin_f32(float(x)))


def arm_cos_f32(x):
    return float(lib().orc_arm_cos_f32(float(x)))


def arm_sin_table():
    return np.ctypeslib.as_array(lib().orc_arm_sin_table(), shape=(513,)).copy()
