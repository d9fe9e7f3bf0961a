"""Pins the CPU oracle: clock-level models vs. stream closed forms, and known
answers derived from the HDL text (the reference ships no vectors of its own)."""
import numpy as np
import pytest

CIC_R, CIC_N = 512, 5


def cic_taps():
    h = np.ones(1, dtype=object)
    box = np.ones(CIC_R, dtype=object)
    for _ in range(CIC_N):
        h = np.convolve(h, box)
    return h  # 2556 taps, python ints


def nco12(orc, phi, n, acc0=0):
    sc, cc, sf, cf = [t.astype(np.int64) for t in orc.nco_tables()]
    acc = (acc0 + phi * np.arange(n, dtype=np.int64)) & 0x3FFFFF
    ic, fi = acc >> 11, acc & 2047
    s28 = sc[ic] * cf[fi] + cc[ic] * sf[fi]
    c28 = cc[ic] * cf[fi] - sc[ic] * sf[fi]
    s14 = np.clip((s28 + 4096) >> 13, -8191, 8191)
    c14 = np.clip((c28 + 4096) >> 13, -8191, 8191)
    return s14 >> 2, c14 >> 2


def wrap(v, bits):
    v = np.asarray(v, dtype=np.int64) & ((1 << bits) - 1)
    return np.where(v >> (bits - 1), v - (1 << bits), v)


def cic_closed_form(orc, adc, phi):
    """y[m] = wrap60(sum_k h[k]*P8[512m-6-k])[59:44], P8 = wrap23(adc*nco12)>>8."""
    n = len(adc)
    s12, c12 = nco12(orc, phi, n)
    outs = []
    h = cic_taps()
    for nco in (s12, c12):
        p8 = wrap(adc.astype(np.int64) * nco, 23) >> 8
        y = []
        for m in range(n // 512):
            top = 512 * m - 6
            acc = 0
            for k in range(len(h)):
                j = top - k
                if j < 0:
                    break
                acc += int(h[k]) * int(p8[j])
            acc &= (1 << 60) - 1
            if acc >> 59:
                acc -= 1 << 60
            y.append(acc >> 44)
        outs.append(np.array(y, dtype=np.int64))
    return outs


def test_nco_tables_formula(orc):
    sc, cc, sf, cf = orc.nco_tables()
    i = np.arange(2048)
    assert np.array_equal(sc, np.round(8191 * np.sin(2 * np.pi * i / 2048)).astype(np.int16))
    assert np.array_equal(cc, np.round(8191 * np.cos(2 * np.pi * i / 2048)).astype(np.int16))
    assert sf.min() == 0 and sf.max() == 25 and np.all(cf == 8191)
    # quarter-wave identities of the coarse ROM
    assert np.array_equal(sc[(i + 512) % 2048], cc)
    assert np.array_equal(cc[(i + 512) % 2048], -sc)


def test_nco_never_saturates(orc):
    """The +-8191 clamp of the output stage is unreachable for all 2^22 phases."""
    sc, cc, sf, cf = [t.astype(np.int64) for t in orc.nco_tables()]
    ic = np.arange(2048)[:, None]
    fi = np.arange(2048)[None, :]
    s = (sc[ic] * cf[fi] + cc[ic] * sf[fi] + 4096) >> 13
    c = (cc[ic] * cf[fi] - sc[ic] * sf[fi] + 4096) >> 13
    assert abs(s).max() <= 8191 and abs(c).max() <= 8191
    for acc in (0, 1, 2047, 2048, 0x100000, 0x2ABCDE, 0x3FFFFF):
        assert orc.nco_eval(acc) == (int(s[acc >> 11, acc & 2047]), int(c[acc >> 11, acc & 2047]))


@pytest.mark.parametrize("phi,seed", [(85333, 1), (605867, 2), (1, 3), (0x3FFFFF, 4)])
def test_rx_front_cycle_model_equals_closed_form(orc, phi, seed):
    rng = np.random.default_rng(seed)
    n = 512 * 9
    adc = rng.integers(-2048, 2048, n).astype(np.int16)
    oi, oq = orc.RxFront(phi).run(adc)
    yi, yq = cic_closed_form(orc, adc, phi)
    assert len(oi) == n // 512
    assert oi[0] == 0 and oq[0] == 0
    assert np.array_equal(oi, yi) and np.array_equal(oq, yq)


def test_rx_front_dc_gain_and_wrap_corner(orc):
    # phi=0 -> sin=0, cos14=8191 -> cos12=2047; constant ADC x gives
    # P8=(x*2047)>>8 and, once the 2556-tap window is full, y = (P8<<45)>>44 = 2*P8
    for x in (2047, -2048, 1000, -1):
        adc = np.full(512 * 8, x, np.int16)
        oi, oq = orc.RxFront(0).run(adc)
        assert np.all(oi == 0)
        assert oq[-1] == 2 * ((x * 2047) >> 8)
    # the only product that overflows 23 bits: (-2048)*(-2048) wraps to -2^22
    s12 = nco12(orc, 0, 1, acc0=0x300000)[0][0]
    assert s12 == -2048
    assert wrap(np.array([-2048 * -2048]), 23)[0] == -(1 << 22)


def test_rx_front_streaming_is_seamless(orc):
    rng = np.random.default_rng(7)
    adc = rng.integers(-2048, 2048, 512 * 12).astype(np.int16)
    a = orc.RxFront(123457)
    whole = a.run(adc)
    b = orc.RxFront(123457)
    parts = [b.run(adc[i:i + 1536]) for i in range(0, len(adc), 1536)]
    assert np.array_equal(whole[0], np.concatenate([p[0] for p in parts]))
    assert np.array_equal(whole[1], np.concatenate([p[1] for p in parts]))


def test_ciccomp_impulse_replays_coefficients(orc):
    from tests.coeffs import RX_CICCOMP_H
    # an impulse of 2^15 would overflow int16; use 2^14 on an even and an odd slot
    for pos in (0, 1):
        x = np.zeros(140, np.int16)
        x[pos] = 1 << 14
        y = orc.RxCicComp().run(x)
        # y[n] = cround(h[2n-pos] * 2^14, 15) = convergent(h/2)
        idx = 2 * np.arange(len(y)) - pos
        hh = np.array([RX_CICCOMP_H[i] if 0 <= i < 65 else 0 for i in idx], dtype=np.int64)
        want = (hh + ((hh >> 1) & 1)) >> 1  # round half to even of h/2
        assert np.array_equal(y, want)


def test_ciccomp_cycle_model_equals_stream_form(orc):
    rng = np.random.default_rng(11)
    x = rng.integers(-32768, 32768, 400).astype(np.int16)
    x[5:9] = -32768
    x[40:44] = 32767
    ys = orc.RxCicComp().run(x)
    yc = orc.rx_ciccomp_cycle(x)
    assert len(yc) >= len(ys) - 1
    assert np.array_equal(yc, ys[:len(yc)])


def test_ciccomp_rounding_half_cases_and_wrap(orc):
    # centre tap 19710 at an even slot: x=-32768 everywhere drives the sum to
    # -sum(h)*32768; wrap16 of the rounded value must follow 31-bit wrap rules
    from tests.coeffs import RX_CICCOMP_H
    x = np.full(200, -32768, np.int16)
    y = orc.RxCicComp().run(x)
    acc = -32768 * int(sum(RX_CICCOMP_H))
    t = ((acc + (1 << 30)) % (1 << 31)) - (1 << 30)
    u = t + (1 << 14) - 1 + ((t >> 15) & 1)
    u = ((u + (1 << 30)) % (1 << 31)) - (1 << 30)
    assert y[-1] == np.int16(u >> 15)


def test_hilbert_impulse_and_antisymmetry(orc):
    from tests.coeffs import RX_HILB_C
    c = np.array(RX_HILB_C, dtype=np.int64)
    assert np.array_equal(c, -c[::-1]) and c[127] == -20824 and c[128] == 20824
    x = np.zeros(300, np.int16)
    x[0] = 1 << 14
    y = orc.RxHilb().run(x)
    prod = c * (1 << 14)
    prod = (prod + ((prod >> 1) & 1)) >> 1          # Q30 -> Q29, convergent
    want = (prod + (1 << 13) - 1 + ((prod >> 14) & 1)) >> 14
    assert np.array_equal(y[:256], want) and np.all(y[256:] == 0)


def test_hilbert_cycle_model_is_stream_form_two_loads_late(orc):
    rng = np.random.default_rng(12)
    x = rng.integers(-32768, 32768, 300).astype(np.int16)
    ys = orc.RxHilb().run(x)
    yc = orc.rx_hilb_cycle(x)
    assert yc[0] == 0 and yc[1] == 0
    assert np.array_equal(yc[2:], ys[:len(yc) - 2])


def test_delay_is_130_samples(orc):
    x = np.arange(1, 400, dtype=np.int16)
    y = orc.Delay130().run(x)
    assert np.all(y[:130] == 0) and np.array_equal(y[130:], x[:-130])


def test_pack12_roundtrip_and_layout(orc):
    x = np.array([-2048, 2047, -1, 0, 0x123, -0x345], np.int16)
    p = orc.pack12(x)
    s0, s1 = 0x800, 0x7FF
    assert list(p[:3]) == [s0 & 0xFF, (s0 >> 8) | ((s1 & 0xF) << 4), s1 >> 4]
    assert np.array_equal(orc.unpack12(p, 6), x)


def test_phrase_from_frequency(orc):
    assert orc.phrase_from_freq(7100000) == (605867, False)
    assert orc.phrase_from_freq(1000000) == (85333, False)
    # second Nyquist zone: folded and I/Q swapped
    ph, sw = orc.phrase_from_freq(30000000)
    assert sw and ph == round((24576000 - (30000000 - 24576000)) / 49152000 * 4194304)
    # third zone: not swapped
    ph, sw = orc.phrase_from_freq(50000000)
    assert not sw and ph == round((50000000 - 49152000) / 49152000 * 4194304)
    # ceil() to the next 100 Hz, as the reference codes it (functions.c:202)
    assert orc.freq_from_phrase(605867) == 7100100


def test_bus_frames(orc):
    assert orc.pack_rx_frame(0x1234, -2, 0x7FFF, -32768) == bytes(
        [0xFF, 0xFE, 0x12, 0x34, 0x80, 0x00, 0x7F, 0xFF])
    assert orc.unpack_tx_frame(bytes([0x80, 0x01, 0x12, 0x34])) == (0x1234, -32767)
    assert orc.parse_params(bytes([0b1100, 0x09, 0x3E, 0xAB])) == (1, 1, 0x093EAB)


def test_window_form_cpu_front_equals_clock_level_model(orc):
    """orc_fast_rx_front_run (the CPU baseline the benchmark times) against the
    clock-level mixer + rx_cic model, from reset: random full scale, the product
    that wraps at every sample, lengths that do not fill the 16 lanes."""
    rng = np.random.default_rng(77)
    cases = [(512 * 40, 605867, None), (512 * 3, 12345, None), (512, 4194303, None), (512 * 17, 3 << 20, -2048),
             (512 * 33, 1, 2047), (1024 * 48, 2097152, None), (512 * 16, 0, -2048)]
    for n, phi, fill in cases:
        x = rng.integers(-2048, 2048, n).astype(np.int16) if fill is None else np.full(n, fill, np.int16)
        wi, wq = orc.RxFront(phi).run(x)
        gi, gq = orc.fast_rx_front(phi, x)
        assert len(wi) == n // 512
        assert np.array_equal(gi, wi) and np.array_equal(gq, wq), (n, phi)


def test_window_form_cpu_chain_equals_clock_level_chain(orc):
    rng = np.random.default_rng(78)
    n, nch = 1024 * 40, 3
    xs = [rng.integers(-2048, 2048, n).astype(np.int16) for _ in range(nch)]
    phis = [605867, 31337, 4000000]
    packed = np.concatenate([orc.pack12(x) for x in xs])
    a = orc.rx_chain_packed(packed, n // 2 * 3, nch, n, phis)
    b = orc.rx_chain_packed(packed, n // 2 * 3, nch, n, phis, fast=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
