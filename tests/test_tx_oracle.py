"""Pins the TX half of the CPU oracle (clock-level models of tx_ciccomp, tx_cic,
tx_mixer, tx_summator, DAC_corrector) and the closed form the GPU kernels use."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import cic_math as cm  # noqa: E402

from tests.coeffs import TX_CICCOMP_P1, TX_CICCOMP_P2  # noqa: E402

H = None


def taps():
    global H
    if H is None:
        H = cm.h_taps()
    return H


def cic_linear(y, t):
    """tx_cic output register after edge t, linear closed form:
    (2^8 * sum_m y[m] h[t - 517 - 512 m]) >> 46."""
    h = taps()
    acc = 0
    for m in range(len(y)):
        k = t - 517 - 512 * m
        if 0 <= k < len(h):
            acc += int(y[m]) * h[k]
    return (acc << 8) >> 46


def test_tx_ciccomp_cycle_model_is_stream_form_one_load_late(orc):
    rng = np.random.default_rng(21)
    x = rng.integers(-32768, 32768, 300).astype(np.int16)
    x[10:40] = 32767
    x[50:80] = -32768
    ys = orc.TxCicComp().run(x)
    yc = orc.tx_ciccomp_cycle(x)
    assert yc[0] == 0 and np.array_equal(yc[1:], ys[:len(yc) - 1])


def test_tx_ciccomp_impulse_replays_both_phases(orc):
    x = np.zeros(30, np.int16)
    x[0] = 1 << 13  # Q14 taps times 2^13, rounded >>14: convergent(h / 2)
    y = orc.TxCicComp().run(x)
    for ph, tab in ((0, TX_CICCOMP_P1), (1, TX_CICCOMP_P2)):
        hh = np.array(tab, dtype=np.int64)
        want = (hh + ((hh >> 1) & 1)) >> 1
        assert np.array_equal(y[ph::2][:24], want)
    assert np.all(y[48:] == 0)


def test_tx_cic_dc_gain_is_one_quarter(orc):
    for v in (1000, -32768, 32767, 4):
        n = 12
        dac, ci, cq = orc.TxBack(0).run(np.full(n, v, np.int16), np.zeros(n, np.int16), want_cic=True)
        assert ci[-1] == v >> 2 and np.all(cq == 0)


@pytest.mark.parametrize("kind", ["random_full_scale", "alternating", "worst_sign", "min_only"])
def test_tx_cic_is_exactly_linear_for_every_16_bit_input(orc, kind):
    """The register widths of tx_cic.vhd equal the worst-case growth, so the
    60-bit wraps of the HDL never trigger and the closed form is exact."""
    rng = np.random.default_rng(22)
    n = 14
    if kind == "random_full_scale":
        y = rng.choice(np.array([-32768, 32767, -1, 0, 12345], dtype=np.int64), n)
    elif kind == "alternating":
        y = np.array([32767 if k % 2 else -32768 for k in range(n)], dtype=np.int64)
    elif kind == "min_only":
        y = np.full(n, -32768, dtype=np.int64)
    else:  # signs of the 4th difference 1,-4,6,-4,1 -> drives the first integrator to its bound
        pat = [-32768, 32767, -32768, 32767, -32768]
        y = np.array([pat[k % 5] for k in range(n)], dtype=np.int64)
    y = y.astype(np.int16)
    dac, ci, cq = orc.TxBack(0).run(y, y[::-1].copy(), want_cic=True)
    for t in list(range(0, 512 * n, 61)) + list(range(512 * 6 - 20, 512 * 6 + 20)):
        assert int(ci[t]) == cic_linear(y, t)
        assert int(cq[t]) == cic_linear(y[::-1], t)


def test_tx_mixer_sum_dac_closed_form_and_no_adder_overflow(orc):
    rng = np.random.default_rng(23)
    n, phi, acc0 = 10, 1234567, 0x2ABCD
    yi = rng.integers(-32768, 32768, n).astype(np.int16)
    yq = rng.integers(-32768, 32768, n).astype(np.int16)
    b = orc.TxBack(phi)
    b.set_phase(acc0)
    dac, ci, cq = b.run(yi, yq, want_cic=True)
    assert not b.overflow()
    for t in range(3, 512 * n, 97):
        s14, c14 = orc.nco_eval((acc0 + (t - 2) * phi) & 0x3FFFFF)
        s = int(ci[t - 3]) * s14 + int(cq[t - 3]) * c14
        assert -(1 << 27) <= s < (1 << 27)
        assert int(dac[t]) == ((s >> 14) + 8191) & 0x3FFF
    assert list(dac[:3]) == [8191, 8191, 8191]
    # the adder cannot overflow: |cic| <= 8192 and |nco| <= 8191
    assert 2 * 8192 * 8191 < (1 << 27)


def test_tx_chain_streaming_is_seamless(orc):
    rng = np.random.default_rng(24)
    iq = rng.integers(-20000, 20000, (12, 2)).astype(np.int16)
    whole = orc.TxChain(605867).run(iq)
    c = orc.TxChain(605867)
    parts = np.concatenate([c.run(iq[:5]), c.run(iq[5:6]), c.run(iq[6:])])
    assert np.array_equal(whole, parts)
