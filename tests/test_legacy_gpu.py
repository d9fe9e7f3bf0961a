"""The firmware-style interface (include/ua3reo_legacy.h) driven the way the
interrupt handlers drive it, against the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests import audio_signals as sig
from tests.legacy_binding import Legacy

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fw():
    import ua3reo_ddc
    lg = Legacy(ua3reo_ddc.lib())
    lg.L.FPGA_Init()
    yield lg
    lg.L.ddc_legacy_shutdown()


def test_rx_ticks_audio_and_fft(fw, orc):
    T = fw.TRX
    T.current_vfo = False
    T.VFO_A.Freq, T.VFO_A.Mode, T.VFO_A.Filter_Width = 7100000, 1, 2700  # USB
    T.AGC, T.DNR, T.NotchFilter, T.Mute, T.CWDecoder = True, False, False, False, False
    T.Agc_speed, T.Volume, T.RF_Gain, T.FFT_Zoom, T.FFT_Averaging, T.SSB_HPF_pass = 3, 50, 50, 1, 4, 300
    T.FFT_Enabled = True
    fw.var("FPGA_NeedSendParams", C.c_bool).value = True
    fw.L.FPGA_fpgadata_stuffclock()
    assert not fw.var("TRX_IQ_swap", C.c_bool).value
    fw.L.initAudioProcessor()
    fw.L.FFT_Init()
    assert fw.err() == ""

    ticks = 512 + 64
    samples = 1024 * ticks
    n = np.arange(samples)
    r = np.random.default_rng(3)
    adc = np.clip(np.rint(800 * np.sin(2 * np.pi * (7100000 + 1300.0) * n / 49152000.0) + r.integers(-30, 30, samples)),
                  -2047, 2047).astype(np.int16)
    assert fw.L.ddc_legacy_adc_push(adc.ctypes.data, samples) == samples
    phi = orc.phrase_from_freq(7100000)[0]
    ws, wv = orc.RxChain(phi).run(adc)
    a = orc.Audio(orc.audio_params(mode="USB", filter_width=2700, hpf_pass=300, agc=1, agc_speed=3, volume=50, rf_gain=50.0,
                                   fft_averaging=4))
    need_rx = fw.var("Processor_NeedRXBuffer", C.c_bool)
    need_fft = fw.var("NeedFFTInputBuffer", C.c_bool)
    index = fw.var("FPGA_Audio_Buffer_Index", C.c_uint16)
    buf_a, buf_b = fw.arr("Processor_AudioBuffer_A", C.c_int32, 384), fw.arr("Processor_AudioBuffer_B", C.c_int32, 384)
    need_fft.value = True  # collect the first 512 frames for the panadapter
    blocks = []
    for t in range(ticks):
        fw.L.FPGA_fpgadata_iqclock()
        # the codec interrupt: one tick after a half buffer has filled
        if index.value in (193, 1) and t >= 192:
            need_rx.value = True
            ready_before = fw.var("Processor_AudioBuffer_ReadyBuffer", C.c_uint8).value
            fw.L.processRxAudio()
            assert not need_rx.value
            blocks.append((buf_b if ready_before == 0 else buf_a).reshape(192, 2).copy())
    assert fw.err() == ""
    assert len(blocks) == 2
    spec_i = fw.arr("FPGA_Audio_Buffer_SPEC_I", C.c_float, 384)
    assert spec_i[5] == float(ws[384 + 5, 0])  # the ring has wrapped once
    want = a.rx(ws[:384], wv[:384])
    assert np.array_equal(np.concatenate(blocks), want)
    assert np.abs(want[192:]).max() > 100
    # panadapter: the first 512 frames were collected while the flag was up
    assert not need_fft.value
    fw.L.FFT_doFFT()
    assert need_fft.value and not fw.var("FFT_need_fft", C.c_bool).value
    bins, row, _ = a.fft(ws[:512])
    got = fw.arr("FFTOutput_mean", C.c_float, 256)
    assert np.abs(got - bins).max() <= 1e-4 * bins.max()
    assert (fw.arr("FFT_waterfall_row", C.c_uint16, 256) != row).sum() <= 2
    assert fw.var("Processor_RX_Audio_Samples_MAX_value", C.c_float).value == a.status()["smeter_max"]


def test_tx_ticks(fw, orc):
    T = fw.TRX
    T.current_vfo = True
    T.VFO_B.Freq, T.VFO_B.Mode, T.VFO_B.Filter_Width = 14150000, 1, 2700
    T.RF_Power, T.Mute, T.SSB_HPF_pass = 60, False, 300
    fw.var("TRX_ptt_hard", C.c_bool).value = True
    fw.var("FPGA_NeedSendParams", C.c_bool).value = True
    fw.L.FPGA_fpgadata_stuffclock()
    fw.L.initAudioProcessor()
    mic = sig.quantise(np.stack([2500 * np.sin(2 * np.pi * 900 * np.arange(192) / 48000.0)] * 2, axis=1))
    buf_a = fw.arr("Processor_AudioBuffer_A", C.c_int32, 384)
    buf_a[:] = mic.reshape(-1)
    fw.var("FPGA_Audio_Buffer_State", C.c_bool).value = False  # fill the first half
    fw.var("Processor_NeedTXBuffer", C.c_bool).value = True
    fw.L.processTxAudio()
    assert fw.err() == "" and not fw.var("Processor_NeedTXBuffer", C.c_bool).value
    a = orc.Audio(orc.audio_params(mode="USB", filter_width=2700, hpf_pass=300, rf_power=60, agc=T.AGC, dnr=T.DNR,
                                   volume=T.Volume, rf_gain=float(T.RF_Gain)))
    wiq, _ = a.tx(mic)
    send_i, send_q = fw.arr("FPGA_Audio_SendBuffer_I", C.c_float, 384), fw.arr("FPGA_Audio_SendBuffer_Q", C.c_float, 384)
    assert np.array_equal(send_i[:192], wiq[:, 0].astype(np.float32))
    assert np.array_equal(send_q[:192], wiq[:, 1].astype(np.float32))
    # eight ticks of command 3 -> 8192 DAC words
    fw.var("FPGA_Audio_Buffer_Index", C.c_uint16).value = 0
    for _ in range(8):
        fw.L.FPGA_fpgadata_iqclock()
    dac = np.zeros(8192, np.uint16)
    assert fw.L.ddc_legacy_dac_pop(dac.ctypes.data, 8192) == 8192
    phi = orc.phrase_from_freq(14150000)[0]
    assert np.array_equal(dac, orc.TxChain(phi).run(wiq[:8]))
    fw.var("TRX_ptt_hard", C.c_bool).value = False


def test_fine_grained_functions(fw, orc):
    T = fw.TRX
    T.current_vfo = False
    T.VFO_A.Mode, T.VFO_A.Filter_Width = 3, 500  # CW_L
    T.AGC, T.Agc_speed, T.CWDecoder = True, 3, True
    fw.L.initAudioProcessor()
    r = np.random.default_rng(8)
    x = r.normal(0, 300, 192).astype(np.float32)
    y = x.copy()
    fw.L.dc_filter(y.ctypes.data, 192, 0)
    a1 = np.float32(1.0 - 2.0 ** -11)
    px, py, want = np.float32(0), np.float32(0), []
    for v in x:
        py = np.float32(np.float32(v - px) + np.float32(a1 * py))
        px = v
        want.append(py)
    assert np.array_equal(y, np.array(want, np.float32))
    g = np.full(192, 100.0, np.float32)
    fw.L.DoAGC(g.ctypes.data, 192)
    # first block: the gain walks from 0 towards 7000 / 100 by 1 / (500 / 3) of the gap, and the
    # block is ramped from the previous applied gain 1.0 to it (agc.c:30-66)
    gain = 70.0 / (500.0 / 3.0)
    assert g[-1] == pytest.approx(100.0 * gain, rel=1e-4)
    assert g[0] == pytest.approx(100.0 * (1.0 - (1.0 - gain) / 192), rel=1e-4)
    nin = r.normal(0, 100, 64).astype(np.float32)
    nout = np.ones(64, np.float32)
    fw.L.processNoiseReduction(nin.ctypes.data, nout.ctypes.data)
    # the weights start at zero and adapt inside the block (mu = 1e-6): first output 0, the rest tiny
    assert nout[0] == 0 and 0 < np.abs(nout).max() < 1e-2
    tone = (500 * np.cos(2 * np.pi * 250.0 * np.arange(192) / 48000.0)).astype(np.float32)
    fw.L.CWDecoder_Process(tone.ctypes.data)
    assert fw.err() == ""


def test_set_mode_policy_and_usb_buffers(fw, orc):
    """TRX_setMode hands the VFO the width of the mode (trx_manager.c:193-220) and
    re-initialises the filters; processRxAudio fills the USB-audio buffer that
    USB_AUDIO_current_rx_buffer names while USB_AUDIO_need_rx_buffer is up
    (audio_processor.c:415-432)."""
    T = fw.TRX
    T.current_vfo = False
    T.SSB_Filter, T.CW_Filter, T.FM_Filter = 3000, 300, 10000
    T.VFO_A.Filter_Width = 1800
    for mode, want in ((1, 3000), (3, 300), (4, 300), (8, 10000), (9, 0), (2, 0), (10, 3000), (0, 3000), (5, 3000)):
        fw.L.TRX_setMode(mode)
        assert fw.err() == ""
        assert (T.VFO_A.Mode, T.VFO_A.Filter_Width) == (mode, want), mode
    T.AGC, T.DNR, T.NotchFilter, T.Mute, T.CWDecoder = False, False, False, False, False
    T.Volume, T.RF_Gain, T.SSB_HPF_pass = 35, 120, 300
    fw.L.TRX_setMode(1)                      # USB, 3000 Hz
    assert fw.err() == ""
    a = orc.Audio(orc.audio_params(mode="USB", filter_width=3000, hpf_pass=300, agc=0, volume=35, rf_gain=120.0))
    need_rx = fw.var("Processor_NeedRXBuffer", C.c_bool)
    need_usb = fw.var("USB_AUDIO_need_rx_buffer", C.c_bool)
    cur_usb = fw.var("USB_AUDIO_current_rx_buffer", C.c_bool)
    index = fw.var("FPGA_Audio_Buffer_Index", C.c_uint16)
    usb = {False: fw.arr("USB_AUDIO_rx_buffer_a", C.c_int16, 384), True: fw.arr("USB_AUDIO_rx_buffer_b", C.c_int16, 384)}
    rings = [fw.arr("FPGA_Audio_Buffer_" + n, C.c_float, 384) for n in ("SPEC_I", "SPEC_Q", "VOICE_I", "VOICE_Q")]
    frames = (np.stack([sig.noisy_two_tone(384, 40), sig.noisy_two_tone(384, 41)]))
    for k, which in enumerate((False, True)):
        blk_s, blk_v = frames[0][192 * k:192 * k + 192], frames[1][192 * k:192 * k + 192]
        rings[0][:192], rings[1][:192] = blk_s[:, 0], blk_s[:, 1]
        rings[2][:192], rings[3][:192] = blk_v[:, 0], blk_v[:, 1]
        index.value = 193                    # the half buffer 0..191 has just filled
        cur_usb.value = which
        usb[False][:] = 77
        usb[True][:] = 77
        need_usb.value = True
        need_rx.value = True
        fw.L.processRxAudio()
        assert fw.err() == "" and not need_usb.value and not need_rx.value
        want = a.rx(blk_s, blk_v)
        assert np.array_equal(usb[which].reshape(192, 2), orc.usb_frames(want, 35))
        assert np.all(usb[not which] == 77)
    # without the flag the buffers stay as they are
    usb[False][:] = 5
    need_rx.value = True
    fw.L.processRxAudio()
    assert np.all(usb[False] == 5)
