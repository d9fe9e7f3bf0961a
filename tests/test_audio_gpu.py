"""Parity of the HIP 48 kHz audio path and FFT panadapter with the CPU oracle,
through the C ABI.

Tolerances.  Everything except the FM discriminator and the FFT is built from
IEEE-754 single-precision add / multiply / divide / sqrt in the firmware's order
with contraction off on both sides, so those modes are compared bit for bit.
That includes the FM modulator, whose sine and cosine are CMSIS table
interpolations on both sides.  NFM / WFM reception goes through atan2f, where the
device math library and libm may differ by a few ulp: outputs must agree within
2e-4 of the block's full scale.  The FFT
runs in float32 on the device and in float64 in the oracle: bins must agree
within 1e-4 of the frame maximum; colours may differ only where a height sits on
a rounding boundary."""
import numpy as np
import pytest

from tests import audio_signals as sig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.lib()
    return ua3reo_ddc


def make_inputs(n_ch, n, seed):
    spec = np.stack([sig.noisy_two_tone(n, seed + 2 * c) for c in range(n_ch)])
    voice = np.stack([sig.noisy_two_tone(n, seed + 2 * c + 1) for c in range(n_ch)])
    return spec, voice


def oracle_run(orc, params_per_ch, spec, voice):
    out, st = [], []
    for c, p in enumerate(params_per_ch):
        a = orc.Audio(p)
        out.append(a.rx(spec[c], voice[c]))
        st.append(a.status())
    return np.stack(out), st


EXACT_CASES = [
    dict(mode="USB"), dict(mode="LSB", hpf_pass=100, filter_width=3000),
    dict(mode="USB", dnr=1), dict(mode="LSB", notch=1, notch_fc=1200),
    dict(mode="CW_U", filter_width=500, cw_decoder=1), dict(mode="CW_L", filter_width=300),
    dict(mode="DIGI_U", filter_width=3000), dict(mode="DIGI_L", filter_width=0),
    dict(mode="AM", filter_width=6000), dict(mode="AM", filter_width=10000, dnr=1, notch=1),
    dict(mode="IQ"), dict(mode="LOOPBACK"), dict(mode="USB", agc=0, iq_swap=1, volume=100),
    dict(mode="USB", mute=1), dict(mode="USB", agc_speed=1, rf_gain=200.0, filter_width=15000),
]


def test_audio_modes_bit_exact(ddc, orc):
    n = 192 * 40
    n_ch = len(EXACT_CASES)
    spec, voice = make_inputs(n_ch, n, 100)
    d = ddc.Ddc(n_ch)
    pg = [ddc.audio_params(**kw) for kw in EXACT_CASES]
    po = [orc.audio_params(**kw) for kw in EXACT_CASES]
    for c in range(n_ch):
        d.audio_configure(pg[c], c)
    got = d.audio_rx_process(spec, voice)
    want, st = oracle_run(orc, po, spec, voice)
    for c in range(n_ch):
        assert np.array_equal(got[c], want[c]), "case %d %r: first diff at %s" % (
            c, EXACT_CASES[c], np.argwhere(got[c] != want[c])[:3])
        g = d.audio_status(c)
        assert g["agc_gain"] == st[c]["agc_gain"]
        assert g["smeter_max"] == st[c]["smeter_max"] and g["smeter_min"] == st[c]["smeter_min"]
        assert g["blocks"] == 40
    assert np.abs(got[0]).max() > 1000


@pytest.mark.parametrize("sizes", [[192] * 6, [192 * 5, 192], [192 * 2, 192 * 3, 192]])
def test_audio_streaming_calls_equal_one_call(ddc, orc, sizes):
    total = sum(sizes)
    spec, voice = make_inputs(3, total, 200)
    cases = [dict(mode="USB", dnr=1, notch=1), dict(mode="AM"), dict(mode="CW_U", filter_width=500)]
    d = ddc.Ddc(3)
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
    outs, pos = [], 0
    for s in sizes:
        outs.append(d.audio_rx_process(spec[:, pos:pos + s], voice[:, pos:pos + s]))
        pos += s
    got = np.concatenate(outs, axis=1)
    want, _ = oracle_run(orc, [orc.audio_params(**kw) for kw in cases], spec, voice)
    assert np.array_equal(got, want)


def test_audio_fm_within_tolerance(ddc, orc):
    n = 192 * 60
    cases = [dict(mode="NFM", filter_width=0), dict(mode="NFM", filter_width=10000, fm_sql_threshold=3),
             dict(mode="WFM", filter_width=0), dict(mode="NFM", agc=0, volume=100, filter_width=0)]
    spec = np.stack([sig.fm_iq(n, 500.0 + 300.0 * c, 2500.0, seed=c) for c in range(4)])
    voice = np.zeros_like(spec)
    d = ddc.Ddc(4)
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
    got = d.audio_rx_process(spec, voice).astype(np.float64)
    want, st = oracle_run(orc, [orc.audio_params(**kw) for kw in cases], spec, voice)
    want = want.astype(np.float64)
    for c in range(4):
        scale = max(np.abs(want[c]).max(), 1.0)
        assert np.abs(got[c] - want[c]).max() <= 2e-4 * scale + 1.0, "case %d" % c
        assert d.audio_status(c)["squelched"] == st[c]["squelched"]
    assert np.abs(want[0]).max() > 100


def test_cw_decoder_text(ddc, orc):
    voice = sig.cw_keyed("SOS SOS CQ DE TEST", wpm=20)[None]
    kw = dict(mode="CW_U", agc=1, cw_decoder=1, rf_gain=10.0, volume=100, filter_width=500)
    d = ddc.Ddc(1)
    d.audio_configure(ddc.audio_params(**kw))
    got = d.audio_rx_process(np.zeros_like(voice), voice)
    a = orc.Audio(orc.audio_params(**kw))
    want = a.rx(np.zeros_like(voice[0]), voice[0])
    assert np.array_equal(got[0], want)
    g, w = d.audio_status(0), a.status()
    assert g["cw_text"] == w["cw_text"] and "CQ DE TEST" in g["cw_text"]
    assert g["cw_wpm"] == w["cw_wpm"]


def test_audio_saturation_and_errors(ddc, orc):
    n = 192
    spec = np.full((1, n, 2), 32767, np.int16)
    spec[:, 1::2] = -32768
    d = ddc.Ddc(1)
    with pytest.raises(ddc.DdcError):
        d.audio_rx_process(spec, spec)  # not configured yet
    with pytest.raises(ddc.DdcError):
        d.audio_configure(ddc.audio_params(filter_width=1234))
    with pytest.raises(ddc.DdcError):
        d.audio_configure(ddc.audio_params(mode="USB", hpf_pass=60))
    d.audio_configure(ddc.audio_params(mode="IQ", rf_gain=1e6, volume=100))
    out = d.audio_rx_process(spec, spec)
    assert out.max() == 2 ** 31 - 1 and out.min() == -2 ** 31
    with pytest.raises(ddc.DdcError):
        d.audio_rx_process(spec[:, :100], spec[:, :100])  # not a multiple of 192


def test_fft_frames_within_tolerance(ddc, orc):
    n_ch, frames = 6, 24
    n = 512 * frames
    r = np.random.default_rng(7)
    spec = np.stack([
        sig.quantise(sig.tone_iq(n, 6000.0, 500.0) + r.normal(0, 5, (n, 2))),
        sig.quantise(sig.tone_iq(n, -9000.0, 3000.0) + r.normal(0, 50, (n, 2))),
        sig.quantise(r.normal(0, 3000.0, (n, 2))),
        sig.noisy_two_tone(n, 41, amp=2000.0),
        sig.quantise(np.concatenate([r.normal(0, 3000.0, (n // 2, 2)), np.zeros((n // 2, 2))])),
        sig.noisy_two_tone(n, 43, amp=100.0),
    ])
    cases = [dict(fft_averaging=1), dict(fft_averaging=4), dict(fft_averaging=2), dict(fft_averaging=3, notch=1),
             dict(fft_averaging=2), dict(fft_averaging=4, mode="LOOPBACK", iq_swap=1)]
    d = ddc.Ddc(n_ch)
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
    # two calls, to carry the auto-level and smoothing state across
    b1, r1 = d.fft_process(spec[:, :512 * 10])
    b2, r2 = d.fft_process(spec[:, 512 * 10:])
    bins, rows = np.concatenate([b1, b2], axis=1), np.concatenate([r1, r2], axis=1)
    mismatched_pixels = 0
    for c in range(n_ch):
        a = orc.Audio(orc.audio_params(**cases[c]))
        for f in range(frames):
            wb, wr, wh = a.fft(spec[c, 512 * f:512 * (f + 1)])
            tol = 1e-4 * max(float(wb.max()), 1e-3)
            assert np.abs(bins[c, f] - wb).max() <= tol, "channel %d frame %d" % (c, f)
            mismatched_pixels += int((rows[c, f] != wr).sum())
    assert mismatched_pixels <= 0.002 * n_ch * frames * 256
    assert (rows[0, -1] >> 11).max() == 31  # something red on the carrier


def test_fft_seventy_channels_equal_one_channel_runs(ddc, orc):
    """Two workgroups of the no-zoom prefilter, the second partly filled, channels with and without
    the notch and the I/Q swap side by side: every checked channel must equal, bit for bit, the
    same channel run alone, and stay within the FFT tolerance of the oracle."""
    n_ch, frames = 70, 3
    n = 512 * frames
    spec = np.stack([sig.noisy_two_tone(n, 900 + c, amp=300.0 + 40.0 * c) for c in range(n_ch)])
    kws = [dict(fft_averaging=1 + c % 4, iq_swap=int(c % 3 == 0), notch=int(c % 5 == 0)) for c in range(n_ch)]
    d = ddc.Ddc(n_ch)
    for c, kw in enumerate(kws):
        d.audio_configure(ddc.audio_params(**kw), c)
    b1, r1 = d.fft_process(spec[:, :512])
    b2, r2 = d.fft_process(spec[:, 512:])
    bins, rows = np.concatenate([b1, b2], axis=1), np.concatenate([r1, r2], axis=1)
    for c in (0, 1, 5, 15, 63, 64, 65, 69):
        one = ddc.Ddc(1)
        one.audio_configure(ddc.audio_params(**kws[c]), 0)
        wb, wr = one.fft_process(spec[c:c + 1])
        assert np.array_equal(bins[c], wb[0]) and np.array_equal(rows[c], wr[0]), "channel %d" % c
        assert np.array_equal(d.fft_waterfall(c), one.fft_waterfall(0)), "channel %d" % c
        one.close()
        a = orc.Audio(orc.audio_params(**kws[c]))
        for f in range(frames):
            ob, _, _ = a.fft(spec[c, 512 * f:512 * (f + 1)])
            assert np.abs(bins[c, f] - ob).max() <= 1e-4 * max(float(ob.max()), 1e-3), "channel %d frame %d" % (c, f)


def mic_signal(n, seed):
    r = np.random.default_rng(seed)
    t = np.arange(n) / sig.FS
    x = 2500.0 * np.sin(2 * np.pi * r.uniform(300, 2500) * t) + 900.0 * np.sin(2 * np.pi * r.uniform(300, 2500) * t)
    x *= 1.0 + 0.5 * np.sin(2 * np.pi * 2.3 * t)
    l = sig.quantise(x + r.normal(0, 20, n))
    rr = sig.quantise(0.5 * x + r.normal(0, 20, n))
    return np.stack([l, rr], axis=1)


TX_EXACT = [
    dict(mode="USB", rf_power=50), dict(mode="LSB", rf_power=100, hpf_pass=100, filter_width=3000),
    dict(mode="DIGI_U", rf_power=30, filter_width=0), dict(mode="DIGI_L", rf_power=70),
    dict(mode="AM", rf_power=50, filter_width=6000), dict(mode="CW_U", rf_power=40, key=1),
    dict(mode="CW_L", rf_power=40, key=0), dict(mode="IQ", rf_power=50), dict(mode="LOOPBACK", volume=70),
    dict(mode="USB", rf_power=50, tune=1), dict(mode="USB", rf_power=50, mute=1),
]


def test_tx_audio_bit_exact(ddc, orc):
    n = 192 * 30
    n_ch = len(TX_EXACT)
    mic = np.stack([mic_signal(n, 300 + c) for c in range(n_ch)])
    d = ddc.Ddc(n_ch)
    for c, kw in enumerate(TX_EXACT):
        d.audio_configure(ddc.audio_params(**kw), c)
    # two calls: the FIR delay lines, lattices and ALC gain carry over
    iq1, lp1 = d.audio_tx_process(mic[:, :192 * 11])
    iq2, lp2 = d.audio_tx_process(mic[:, 192 * 11:])
    iq, lp = np.concatenate([iq1, iq2], axis=1), np.concatenate([lp1, lp2], axis=1)
    for c, kw in enumerate(TX_EXACT):
        a = orc.Audio(orc.audio_params(**kw))
        wiq, wlp = a.tx(mic[c])
        assert np.array_equal(iq[c], wiq), "case %d %r: first diff at %s" % (c, kw, np.argwhere(iq[c] != wiq)[:3])
        assert np.array_equal(lp[c], wlp), "case %d %r loopback" % (c, kw)
        assert d.audio_status(c)["alc_gain"] == a.alc_gain()
    assert np.abs(iq[0]).max() > 1000


def test_tx_fm_bit_exact(ddc, orc):
    # the modulator's sine / cosine are CMSIS table interpolations (float multiplies and adds only)
    # on both sides, so the 16-bit bus words must be equal
    n = 192 * 30
    cases = [dict(mode="NFM", rf_power=60, filter_width=0), dict(mode="WFM", rf_power=100, filter_width=15000),
             dict(mode="NFM", rf_power=20, filter_width=8000)]
    mic = np.stack([mic_signal(n, 400 + c) for c in range(3)])
    d = ddc.Ddc(3)
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
    iq, _ = d.audio_tx_process(mic)
    for c, kw in enumerate(cases):
        wiq, _ = orc.Audio(orc.audio_params(**kw)).tx(mic[c])
        assert np.array_equal(iq[c], wiq), "case %d" % c
        assert np.abs(iq[c].astype(np.int32)).max() > 1000


def test_tx_device_chain_mic_to_dac(ddc, orc):
    """The whole transmit direction on device pointers: microphone frames -> processTxAudio
    (ddc_audio_tx_process_dev) -> the FPGA's TX chain (ddc_tx_process_dev) -> DAC words,
    two calls in a row, against the oracle's two halves chained the same way."""
    torch = pytest.importorskip("torch")
    n, n_ch = 192 * 2, 3
    cases = [dict(mode="USB", rf_power=50), dict(mode="NFM", rf_power=80, filter_width=0), dict(mode="AM", rf_power=40)]
    phis = [ddc.phrase_from_frequency(f)[0] for f in (7100000, 14200000, 3650000)]
    mic = np.stack([mic_signal(2 * n, 800 + c) for c in range(n_ch)])
    d = ddc.Ddc(n_ch)
    d.tune_many(phis)
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
    mic_d = torch.from_numpy(mic).cuda()
    iq_d = torch.zeros((n_ch, n, 2), dtype=torch.int16, device="cuda")
    dac_d = torch.zeros((2, n_ch, 1024 * n), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for k in range(2):
        part = mic_d[:, k * n:(k + 1) * n].contiguous()
        torch.cuda.synchronize()
        d.audio_tx_process_dev(part.data_ptr(), n, iq_d.data_ptr())
        d.tx_process_dev(iq_d.data_ptr(), n, dac_d[k].data_ptr())
        d.sync()
    dac = dac_d.cpu().numpy().view(np.uint16)
    for c, kw in enumerate(cases):
        wiq, _ = orc.Audio(orc.audio_params(**kw)).tx(mic[c])
        want = orc.TxChain(phis[c]).run(wiq)
        assert np.array_equal(np.concatenate([dac[0, c], dac[1, c]]), want), "case %d" % c
        assert len(set(want.tolist())) > 100


def test_audio_stage_runs_single_firmware_stages(ddc, orc):
    """ddc_audio_stage, the entry the firmware-style shim is built on: the DC filter and the AGC
    as stages of their own must equal the same stages inside a full processRxAudio run of the
    oracle only where nothing else touches the samples, so they are checked by their own laws:
    the DC filter's recursion restated here, and the AGC's block gain never exceeding its limits."""
    d = ddc.Ddc(1)
    d.audio_configure(ddc.audio_params(mode="USB", agc=1))
    r = np.random.default_rng(5)
    x = (r.normal(0, 3000.0, 192) + 700.0).astype(np.float32)
    y = d.audio_stage(0, 0, 0, x)                         # dc_filter, state slot 0
    want = np.zeros(192, np.float32)
    px, py = np.float32(0), np.float32(0)
    a1 = np.float32(0.99951171875)
    for i in range(192):                                  # audio_filters.c:358-373
        o = np.float32(np.float32(x[i] - px) + np.float32(a1 * py))
        px, py = x[i], o
        want[i] = o
    assert np.array_equal(y, want)
    y2 = d.audio_stage(0, 0, 0, x)                        # the state carries over
    assert not np.array_equal(y2, y)
    g = d.audio_stage(0, 1, 0, x)                         # DoAGC on the block
    assert np.isfinite(g).all() and np.abs(g).max() <= 10000.0 * 1.0001
    with pytest.raises(ddc.DdcError):
        d.audio_stage(0, 9, 0, x)


@pytest.mark.parametrize("zoom", [2, 4, 8, 16])
def test_fft_zoom_within_tolerance(ddc, orc, zoom):
    n_ch, frames = 3, 20
    n = 512 * frames
    r = np.random.default_rng(17 + zoom)
    spec = np.stack([
        sig.quantise(sig.tone_iq(n, 700.0, 900.0) + r.normal(0, 10, (n, 2))),
        sig.quantise(r.normal(0, 2000.0, (n, 2))),
        sig.noisy_two_tone(n, 47, amp=1500.0),
    ])
    kw = dict(fft_zoom=zoom, fft_averaging=2)
    d = ddc.Ddc(n_ch)
    d.audio_configure(ddc.audio_params(**kw))
    b1, r1 = d.fft_process(spec[:, :512 * 7])
    b2, r2 = d.fft_process(spec[:, 512 * 7:])
    bins, rows = np.concatenate([b1, b2], axis=1), np.concatenate([r1, r2], axis=1)
    bad = 0
    for c in range(n_ch):
        a = orc.Audio(orc.audio_params(**kw))
        for f in range(frames):
            wb, wr, _ = a.fft(spec[c, 512 * f:512 * (f + 1)])
            assert np.abs(bins[c, f] - wb).max() <= 1e-4 * max(float(wb.max()), 1e-3), "ch %d frame %d" % (c, f)
            bad += int((rows[c, f] != wr).sum())
    assert bad <= 0.002 * n_ch * frames * 256


def test_fft_mixed_zoom_is_rejected(ddc):
    d = ddc.Ddc(2)
    d.audio_configure(ddc.audio_params(fft_zoom=1), 0)
    d.audio_configure(ddc.audio_params(fft_zoom=4), 1)
    with pytest.raises(ddc.DdcError):
        d.fft_process(np.zeros((2, 512, 2), np.int16))
    with pytest.raises(ddc.DdcError):
        d.audio_configure(ddc.audio_params(fft_zoom=3))


def test_full_chain_adc_to_audio(ddc, orc):
    """Config-2 shape at test size: packed ADC -> DDC -> USB demodulation, and the
    TX return path mic -> I/Q -> DUC, every stage against the oracle."""
    samples = 1024 * 192 * 3
    freq = 7100000
    phi, _ = ddc.phrase_from_frequency(freq)
    n = np.arange(samples)
    r = np.random.default_rng(5)
    adc = np.clip(np.rint(900 * np.sin(2 * np.pi * (freq + 1500.0) * n / 49152000.0) +
                          r.integers(-40, 40, samples)), -2047, 2047).astype(np.int16)
    d = ddc.Ddc(1)
    d.tune(0, phi)
    kw = dict(mode="USB", filter_width=2700, hpf_pass=300, agc=1, dnr=0)
    d.audio_configure(ddc.audio_params(**kw))
    spec, voice = d.rx_process(ddc.pack_adc12(adc)[None], samples)
    audio = d.audio_rx_process(spec, voice)
    ws, wv = orc.RxChain(phi).run(adc)
    assert np.array_equal(spec[0], ws) and np.array_equal(voice[0], wv)
    a = orc.Audio(orc.audio_params(**kw))
    assert np.array_equal(audio[0], a.rx(ws, wv))
    assert np.abs(audio[0][192:]).max() > 100
    # TX: two blocks of microphone -> I/Q -> DAC words
    mic = mic_signal(192 * 2, 9)[None]
    iq, _ = d.audio_tx_process(mic)
    wiq, _ = a.tx(mic[0])
    assert np.array_equal(iq[0], wiq)
    dac = d.tx_process(iq[:, :8])
    assert np.array_equal(dac[0], orc.TxChain(phi).run(wiq[:8]))


def test_waterfall_and_retune(ddc, orc):
    frames = 57
    n = 512 * frames
    r = np.random.default_rng(23)
    spec = np.stack([sig.quantise(sig.tone_iq(n, 6000.0, 500.0) + r.normal(0, 20, (n, 2))),
                     sig.noisy_two_tone(n, 51, amp=1500.0)])
    cases = [dict(fft_averaging=2, fft_zoom=2), dict(fft_averaging=3, fft_zoom=2)]
    d = ddc.Ddc(2)
    oracles = []
    for c, kw in enumerate(cases):
        d.audio_configure(ddc.audio_params(**kw), c)
        oracles.append(orc.Audio(orc.audio_params(**kw)))
    # three calls of different length, a retune in between
    cuts = [0, 512 * 3, 512 * 55, n]
    for k in range(3):
        d.fft_process(spec[:, cuts[k]:cuts[k + 1]])
        for c in range(2):
            for f in range(cuts[k] // 512, cuts[k + 1] // 512):
                oracles[c].fft(spec[c, 512 * f:512 * (f + 1)])
        if k == 1:
            for c, hz in enumerate((1000, -2500)):
                d.fft_move_waterfall(c, hz)
                oracles[c].move_waterfall(hz)
    for c in range(2):
        got, want = d.fft_waterfall(c), oracles[c].waterfall()
        assert (got != want).sum() <= 0.002 * got.size, "channel %d" % c
        assert want.any()
    with pytest.raises(ddc.DdcError):
        d.fft_move_waterfall(0, 40000)


def test_smeter_dbm_matches_oracle(ddc, orc):
    n = 192 * 6
    spec, voice = make_inputs(2, n, 700)
    d = ddc.Ddc(2)
    kws = [dict(mode="USB", rf_gain=20.0), dict(mode="AM", rf_gain=3.0)]
    for c, kw in enumerate(kws):
        d.audio_configure(ddc.audio_params(**kw), c)
    d.audio_rx_process(spec, voice)
    for c, kw in enumerate(kws):
        a = orc.Audio(orc.audio_params(**kw))
        a.rx(spec[c], voice[c])
        assert d.smeter_dbm(c) == a.smeter_dbm()
        assert d.audio_status(c)["smeter_max"] == 0.0


def test_usb_audio_frames_match_oracle(ddc, orc):
    """ddc_audio_usb_frames(_dev) (audio_processor.c:415-432) on the codec frames of a real
    RX block and on hand-made extremes, one volume per channel incl. 0, 1, 100 and 255."""
    torch = pytest.importorskip("torch")
    vols = [50, 1, 100, 0, 255, 20, 33]
    n_ch, n = len(vols), 192 * 6
    spec, voice = make_inputs(n_ch, n, 900)
    d = ddc.Ddc(n_ch)
    for c, v in enumerate(vols):
        d.audio_configure(ddc.audio_params(mode="USB", volume=v, agc=0, rf_gain=250.0), channel=c)
    audio = d.audio_rx_process(spec, voice)
    rng = np.random.default_rng(5)
    audio[:, :64] = rng.integers(-2 ** 31, 2 ** 31, size=(n_ch, 64, 2), dtype=np.int64).astype(np.int32)
    audio[:, 64:70] = np.array([[2 ** 31 - 1, -2 ** 31], [0, 1], [-1, 16383], [32767, -32768], [65535, 40000],
                                [-40000, 7]], np.int32)
    got = d.audio_usb_frames(audio)
    for c, v in enumerate(vols):
        assert np.array_equal(got[c], orc.usb_frames(audio[c], v)), "volume %d" % v
    # device-pointer form, after a volume change through ddc_audio_update
    d.audio_update(ddc.audio_params(mode="USB", volume=10, agc=0, rf_gain=250.0), channel=0)
    a_dev = torch.from_numpy(audio).cuda()
    u_dev = torch.zeros((n_ch, n, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    d.audio_usb_frames_dev(a_dev.data_ptr(), n, u_dev.data_ptr())
    d.sync()
    assert np.array_equal(u_dev[0].cpu().numpy(), orc.usb_frames(audio[0], 10))
    assert np.array_equal(u_dev[1:].cpu().numpy(), got[1:])
    d.close()


def test_config3_chain_256_channels_ddc_into_panadapter(ddc, orc):
    """BASELINE config 3 at its shape: 256 tuned channels x 2^22 ADC samples through
    ddc_rx_process_dev -> ddc_fft_process_dev on device buffers (eight panadapter rows per
    channel).  One call equals two half calls for every channel, bit for bit, and
    channels 0, 127 and 255 match the oracle chain (SPEC exactly, bins within 1e-4 of the
    frame maximum, colour rows up to boundary pixels)."""
    torch = pytest.importorskip("torch")
    n_ch, samples = 256, 1 << 22
    rb = samples // 2 * 3
    nf = samples // 1024
    probe = (0, 127, 255)
    g = torch.Generator(device="cuda").manual_seed(303)
    adc = torch.randint(0, 256, (n_ch, rb), dtype=torch.uint8, device="cuda", generator=g)
    host = {c: ddc.synth_adc(c, samples) for c in probe}
    for c in probe:
        adc[c] = torch.from_numpy(ddc.pack_adc12(host[c])).cuda()
    phis = [ddc.phrase_from_frequency(1000000 + 48000 * c)[0] for c in range(n_ch)]
    params = dict(mode="USB", fft_averaging=4)

    def run(parts):
        d = ddc.Ddc(n_ch)
        d.tune_many(phis)
        d.audio_configure(ddc.audio_params(**params))
        spec = torch.zeros((n_ch, nf, 2), dtype=torch.int16, device="cuda")
        voice = torch.zeros_like(spec)
        bins = torch.zeros((n_ch, nf // 512, 256), dtype=torch.float32, device="cuda")
        rows = torch.zeros((n_ch, nf // 512, 256), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        step, fstep = samples // parts, nf // parts
        for k in range(parts):
            # each part's outputs are compact [channel][frames of the part]: stage them, then place them
            s_k = torch.zeros((n_ch, fstep, 2), dtype=torch.int16, device="cuda")
            v_k = torch.zeros_like(s_k)
            b_k = torch.zeros((n_ch, fstep // 512, 256), dtype=torch.float32, device="cuda")
            r_k = torch.zeros((n_ch, fstep // 512, 256), dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()
            d.rx_process_dev(adc.data_ptr() + k * (step // 2 * 3), rb, step, s_k.data_ptr(), v_k.data_ptr())
            d.fft_process_dev(s_k.data_ptr(), fstep, b_k.data_ptr(), r_k.data_ptr())
            d.sync()
            spec[:, k * fstep:(k + 1) * fstep] = s_k
            bins[:, k * (fstep // 512):(k + 1) * (fstep // 512)] = b_k
            rows[:, k * (fstep // 512):(k + 1) * (fstep // 512)] = r_k
        d.close()
        return spec.cpu().numpy(), bins.cpu().numpy(), rows.cpu().numpy().view(np.uint16)

    s1, b1, r1 = run(1)
    s2, b2, r2 = run(2)
    assert np.array_equal(s1, s2) and np.array_equal(b1, b2) and np.array_equal(r1, r2)
    assert np.isfinite(b1).all() and b1.max() > 0
    bad_pixels = 0
    for c in probe:
        ws, _ = orc.RxChain(phis[c]).run(host[c])
        assert np.array_equal(s1[c], ws), c
        a = orc.Audio(orc.audio_params(**params))
        for f in range(nf // 512):
            wb, wr, _ = a.fft(ws[512 * f:512 * (f + 1)])
            assert np.abs(b1[c, f] - wb).max() <= 1e-4 * max(float(wb.max()), 1e-3), (c, f)
            bad_pixels += int((r1[c, f] != wr).sum())
    assert bad_pixels <= 12   # of 3 channels x 8 rows x 256 pixels: bins that sit on a colour boundary


def test_config2_one_channel_one_second_to_audio(ddc, orc):
    """BASELINE config 2 at its shape: one channel, one second of a 65 MS/s converter
    (347 x 192 x 1024 = 68.2 M samples) through NCO + CIC + compensator + Hilbert and the
    USB demodulator (2.7 kHz LPF, 300 Hz HPF, AGC on, DNR off) to 66624 audio frames at
    65 MHz / 1024.  SPEC / VOICE and the audio frames are compared with the oracle bit for
    bit over the whole second; the GPU side runs on device buffers."""
    torch = pytest.importorskip("torch")
    fs = 65000000
    samples = 347 * 192 * 1024
    nf = samples // 1024
    rb = samples // 2 * 3
    d = ddc.Ddc(1, fs_adc=fs)
    phi = orc.phrase_from_freq(1000000, fs)[0]
    assert not d.tune_hz(0, 1000000)             # first Nyquist zone: no I/Q swap
    assert ddc.phrase_from_frequency(1000000, fs)[0] == phi
    params = dict(mode="USB", filter_width=2700, hpf_pass=300, agc=1, dnr=0)
    d.audio_configure(ddc.audio_params(**params))
    adc = torch.empty((1, rb), dtype=torch.uint8, device="cuda")
    spec = torch.zeros((1, nf, 2), dtype=torch.int16, device="cuda")
    voice = torch.zeros_like(spec)
    audio = torch.zeros((1, nf, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ddc.synth_adc_dev(0, 1, samples, adc.data_ptr(), rb, fs)
    d.rx_process_dev(adc.data_ptr(), rb, samples, spec.data_ptr(), voice.data_ptr())
    d.audio_rx_process_dev(spec.data_ptr(), voice.data_ptr(), nf, audio.data_ptr())
    d.sync()
    x = ddc.unpack_adc12(adc[0].cpu().numpy(), samples)
    assert np.array_equal(x[:4096], ddc.synth_adc(0, 4096, fs))
    ws, wv = orc.RxChain(phi).run(x)
    assert np.array_equal(spec[0].cpu().numpy(), ws) and np.array_equal(voice[0].cpu().numpy(), wv)
    want = orc.Audio(orc.audio_params(**params)).rx(ws, wv)
    got = audio[0].cpu().numpy()
    assert np.array_equal(got, want)
    assert np.abs(got[nf // 2:]).max() > 1000      # the 1 kHz beat of the two tones is in the passband
    d.close()
