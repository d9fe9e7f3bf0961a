"""The committed fixtures (tests/golden/ddc_golden.npz, written by
tests/golden/make_golden.py) against the oracle, so that the oracle cannot drift
silently, plus checks of the fixtures themselves against the reference's tables."""
import ast
import os

import numpy as np
import pytest

from tests.coeffs import RX_CICCOMP_H, RX_HILB_C

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddc_golden.npz"), allow_pickle=False)
RX_CASES = ["impulse", "step", "dc_pos_fullscale", "dc_neg_fullscale", "tone_fs4", "tone_near_lo", "lfsr"]


def cround(v, s):
    """round half to even of v / 2^s"""
    q, r = divmod(int(v), 1 << s)
    half = 1 << (s - 1)
    return q + (1 if (r > half or (r == half and (q & 1))) else 0)


@pytest.mark.parametrize("name", RX_CASES)
def test_rx_chain_matches_golden(orc, name):
    adc, phi = G["rx_%s_adc" % name], int(G["rx_%s_phi" % name])
    ci, cq = orc.RxFront(phi).run(adc)
    assert np.array_equal(np.stack([ci, cq], axis=1), G["rx_%s_cic" % name])
    spec, voice = orc.RxChain(phi).run(adc)
    assert np.array_equal(spec, G["rx_%s_spec" % name]) and np.array_equal(voice, G["rx_%s_voice" % name])
    assert len(G["rx_%s_cic" % name]) >= 8


def test_golden_dc_gain_of_the_cic():
    # full-scale DC through cos = 2047 (phi = 0 would be; here phi = 1 is 0.01 Hz): the CIC
    # settles to (2047 * 2047 >> 8) * 512^5 >> 44 on the cosine branch
    cic = G["rx_dc_pos_fullscale_cic"]
    want = (((2047 * 2047) & 0x7FFFFF) >> 8) * 512 ** 5 >> 44
    assert abs(int(cic[-1, 1]) - want) <= 1
    # the sine branch only sees the 0.018 rad the NCO has turned by then
    assert 0 < int(cic[-1, 0]) < 700


def test_golden_ciccomp_impulse_replays_the_taps():
    out = G["ciccomp_impulse_out"]
    # y[n] = cround(h[2n] * 16384, 15) = cround(h[2n], 1)
    want = [cround(RX_CICCOMP_H[2 * n], 1) for n in range(33)]
    assert list(out[:33]) == want and not out[33:].any()


def test_golden_hilbert_impulse_replays_the_taps():
    out = G["hilb_impulse_out"]
    # per-product rounding by 1 bit, then 14 more: c * 16384 -> cround(cround(c * 16384, 1), 14)
    want = [cround(cround(int(c) * 16384, 1), 14) for c in RX_HILB_C]
    assert list(out[:256]) == want and not out[256:].any()


def test_stage_fixtures_match_oracle(orc):
    for k in ("impulse", "random", "negfs"):
        assert np.array_equal(orc.RxCicComp().run(G["ciccomp_%s_in" % k]), G["ciccomp_%s_out" % k])
    for k in ("impulse", "random"):
        assert np.array_equal(orc.RxHilb().run(G["hilb_%s_in" % k]), G["hilb_%s_out" % k])
    for k in ("impulse", "dc_fullscale", "neg_fullscale", "random"):
        assert np.array_equal(orc.TxChain(int(G["tx_%s_phi" % k])).run(G["tx_%s_iq" % k]), G["tx_%s_dac" % k])


def test_golden_tx_dc_level():
    # full-scale I = Q DC, 12 pairs in: the compensator (group delay 12 pairs) is still
    # rising, and the DAC already swings around mid scale 8191
    dac = G["tx_dc_fullscale_dac"].astype(np.int64)
    tail = dac[-4096:]
    assert tail.max() <= 0x3FFF and abs(tail.mean() - 8191) < 200 and tail.max() - tail.min() > 1000


def test_phrase_table(orc):
    for f, w, sw in zip(G["phrase_freq"], G["phrase_word"], G["phrase_swap"]):
        assert orc.phrase_from_freq(int(f)) == (int(w), bool(sw))
    tab = dict(zip(G["phrase_freq"].tolist(), zip(G["phrase_word"].tolist(), G["phrase_swap"].tolist())))
    assert tab[7100000] == (605867, 0)          # STM32/Src/functions.c:206-226 by hand
    assert tab[30000000][1] == 1                 # second Nyquist zone: folded, I/Q swapped
    assert tab[30000000][0] == orc.phrase_from_freq(49152000 - 30000000)[0]


def test_frames_bytes(orc):
    spec, voice, raw = G["frames_spec"], G["frames_voice"], G["frames_rx_bytes"]
    for k in range(16):
        b = raw[8 * k:8 * k + 8]
        want = []
        for v in (spec[k, 1], spec[k, 0], voice[k, 1], voice[k, 0]):  # Q, I, Q, I big endian
            want += [(int(v) >> 8) & 0xFF, int(v) & 0xFF]
        assert list(b) == want


def test_audio_fixtures_match_oracle(orc):
    for i in range(5):
        kw = dict(ast.literal_eval(str(G["audio%d_kw" % i])))
        a = orc.Audio(orc.audio_params(**kw))
        assert np.array_equal(a.rx(G["audio%d_spec" % i], G["audio%d_voice" % i]), G["audio%d_out" % i])
    for i in range(2):
        kw = dict(ast.literal_eval(str(G["txaudio%d_kw" % i])))
        a = orc.Audio(orc.audio_params(**kw))
        assert np.array_equal(a.tx(G["txaudio%d_mic" % i])[0], G["txaudio%d_iq" % i])
    a = orc.Audio(orc.audio_params(fft_averaging=2))
    for f in range(6):
        b, r, _ = a.fft(G["fft_spec"][512 * f:512 * (f + 1)])
        assert np.array_equal(b, G["fft_bins"][f]) and np.array_equal(r, G["fft_rows"][f])
