"""Pins the constant tables of both sides to the reference's own numbers.

tests/golden/ref_tables.npz holds the taps, lattice tables, zoom-FFT coefficients
and the four NCO ROM images as tests/golden/make_ref_tables.py read them out of the
reference's HDL / C text and .hex files.  The HIP library reports what is compiled
into it through ddc_debug_table (host call, no GPU); the oracle has a copy of its
own (oracle/oracle_tables.h, oracle/gen_tables.py).  A tap that is wrong on either
side turns one of these tests red; a tap wrong on one side only also breaks the
GPU-vs-oracle parity tests, because the two sides share no table source.

This is the only pin the reference offers (it ships no vectors): the arithmetic
around the tables stays "parity unpinned"."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = np.load(os.path.join(ROOT, "tests", "golden", "ref_tables.npz"))

LPF_NAMES = ["0k3", "0k5", "1k4", "1k6", "1k8", "2k1", "2k3", "2k5", "2k7", "2k9", "3k0", "3k2", "3k4", "3k6",
             "3k8", "4k0", "4k2", "4k4", "4k6", "4k8", "5k0", "5k5", "6k0", "6k5", "7k0", "7k5", "8k0", "8k5",
             "9k0", "9k5", "10k0", "15k0"]
HPF_HZ = [60, 100, 200, 300, 400, 500]


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.lib()
    return ua3reo_ddc


def _rx_proto():
    p1, p2 = REF["rx_ciccomp_phase1"], REF["rx_ciccomp_phase2"]
    h = np.zeros(65, np.int16)
    h[0::2] = p1
    h[1::2] = p2[:32]
    assert p2[32] == 0
    return h


def test_library_integer_taps_are_the_reference_taps(ddc):
    assert np.array_equal(ddc.debug_table("rx_ciccomp_h", np.int16), _rx_proto())
    assert np.array_equal(ddc.debug_table("rx_hilb_c", np.int16), REF["rx_hilb_coeff"])
    assert np.array_equal(ddc.debug_table("tx_ciccomp_p1", np.int16), REF["tx_ciccomp_phase1"])
    assert np.array_equal(ddc.debug_table("tx_ciccomp_p2", np.int16), REF["tx_ciccomp_phase2"])


def test_library_nco_tables_are_the_reference_roms(ddc):
    """The coarse table uploaded for the kernels holds the two coarse ROM images, and
    the kernels' arithmetic fine sine equals the fine sine ROM for all 2048 phases; the
    fine cosine ROM is 8191 throughout, which is the constant the kernels multiply by."""
    t = ddc.debug_table("nco_coarse", np.uint32).reshape(2048, 2)
    sc, cc = REF["nco_sin_c"].astype(np.int64), REF["nco_cos_c"].astype(np.int64)
    lo = lambda w: ((w & 0xFFFF) ^ 0x8000).astype(np.int64) - 0x8000
    hi = lambda w: ((w >> 16) ^ 0x8000).astype(np.int64) - 0x8000
    assert np.array_equal(lo(t[:, 0]), 2 * cc) and np.array_equal(hi(t[:, 0]), 2 * sc)
    assert np.array_equal(lo(t[:, 1]), -2 * sc) and np.array_equal(hi(t[:, 1]), 2 * cc)
    assert np.array_equal(ddc.debug_table("nco_fine_sine", np.int16), REF["nco_sin_f"])
    assert np.all(REF["nco_cos_f"] == 8191)


def test_library_audio_tables_are_the_reference_tables(ddc):
    f32 = lambda a: np.asarray(a, np.float64).astype(np.float32)
    assert np.array_equal(ddc.debug_table("tx_hilbert_i", np.float32), f32(REF["af_i_tx_coeffs"]))
    assert np.array_equal(ddc.debug_table("tx_hilbert_q", np.float32), f32(REF["af_q_tx_coeffs"]))
    assert list(ddc.debug_table("hpf_hz", np.int32)) == HPF_HZ
    hk = ddc.debug_table("hpf_k", np.float32).reshape(6, 6)
    hv = ddc.debug_table("hpf_v", np.float32).reshape(6, 7)
    for i, f in enumerate(HPF_HZ):
        assert np.array_equal(hk[i], f32(REF["af_IIR_HPF_%d_PKcoeffs" % f]))
        assert np.array_equal(hv[i], f32(REF["af_IIR_HPF_%d_PVcoeffs" % f]))
    lk = ddc.debug_table("lpf_k", np.float32).reshape(32, 11)
    lv = ddc.debug_table("lpf_v", np.float32).reshape(32, 12)
    st = ddc.debug_table("lpf_stages", np.int32)
    hz = ddc.debug_table("lpf_hz", np.int32)
    for i, n in enumerate(LPF_NAMES):
        k, v = f32(REF["af_IIR_LPF_%s_PKcoeffs" % n]), f32(REF["af_IIR_LPF_%s_PVcoeffs" % n])
        assert st[i] == len(k) and len(v) == len(k) + 1
        assert hz[i] == int(round(float(n.replace("k", ".")) * 1000))
        assert np.array_equal(lk[i, :len(k)], k) and not lk[i, len(k):].any()
        assert np.array_equal(lv[i, :len(v)], v) and not lv[i, len(v):].any()
    assert np.array_equal(ddc.debug_table("sql_hpf_k", np.float32), f32(REF["af_IIR_HPF_15k0_PKcoeffs"]))
    assert np.array_equal(ddc.debug_table("sql_hpf_v", np.float32), f32(REF["af_IIR_HPF_15k0_PVcoeffs"]))
    assert np.array_equal(ddc.debug_table("zoom_biquad", np.float64).reshape(4, 20), REF["fft_zoom_biquad"])
    assert np.array_equal(ddc.debug_table("zoom_fir", np.float64).reshape(4, 4), REF["fft_zoom_fir"])


def test_unknown_table_is_refused(ddc):
    with pytest.raises(ddc.DdcError):
        ddc.debug_table("no_such_table", np.int16)


def test_oracle_nco_roms_are_the_reference_roms(orc):
    sc, cc, sf, cf = orc.nco_tables()
    assert np.array_equal(sc, REF["nco_sin_c"]) and np.array_equal(cc, REF["nco_cos_c"])
    assert np.array_equal(sf, REF["nco_sin_f"]) and np.array_equal(cf, REF["nco_cos_f"])


def test_oracle_impulse_responses_replay_the_reference_taps(orc):
    """The oracle's own tables, read back through its arithmetic: an impulse of 2^14 into
    rx_ciccomp (Q15, >> 15 with rounding) gives h / 2 rounded half to even; an impulse into
    the TX compensator (Q14) at full scale gives the taps themselves."""
    x = np.zeros(140, np.int16)
    x[0] = 1 << 14
    y = orc.RxCicComp().run(x)            # outputs at even input indices: taps 0, 2, 4, ...
    h = _rx_proto().astype(np.int64)
    want = [(int(v) + 1) >> 1 if (int(v) & 3) == 3 else int(v) >> 1 for v in h[0::2]]
    assert list(y[:33]) == want
    x = np.zeros(300, np.int16)
    x[0] = 1 << 14
    c = REF["rx_hilb_coeff"].astype(np.int64)
    # per-product round to Q29, then >> 14 with rounding: c * 2^14 / 2 / 2^14 = c / 2 rounded half to even
    want = [(int(v) + 1) >> 1 if (int(v) & 3) == 3 else int(v) >> 1 for v in c]
    assert list(orc.RxHilb().run(x)[:256]) == want
    x = np.zeros(30, np.int16)
    x[0] = 1 << 14
    y = orc.TxCicComp().run(x)            # interleaved phase 1 / phase 2 outputs
    assert list(y[0:48:2]) == [int(v) for v in REF["tx_ciccomp_phase1"]]
    assert list(y[1:48:2]) == [int(v) for v in REF["tx_ciccomp_phase2"]]


def test_arm_sin_table_and_interpolation_on_both_sides(ddc, orc):
    """arm_sin_f32 / arm_cos_f32 are CMSIS-DSP routines (1.6.0, not part of the reference tree):
    parity unpinned.  What can be held: both sides build the same 513-entry table (the float of
    the 8-decimal literal of sin(2 pi i / 512)) from their own generators, the table has the
    library's landmarks, and the interpolated value stays within the table's interpolation
    error of the true sine and is exact at table points."""
    lib_t, orc_t = ddc.debug_table("arm_sin", np.float32), orc.arm_sin_table()
    assert lib_t.shape == (513,) and np.array_equal(lib_t.view(np.uint32), orc_t.view(np.uint32))
    assert lib_t[0] == 0 and lib_t[128] == 1 and lib_t[384] == -1 and lib_t[256] == 0
    assert np.signbit(lib_t[512]) and lib_t[512] == 0          # -0.00000000f
    assert lib_t[1] == np.float32(0.01227154) and lib_t[511] == np.float32(-0.01227154)
    true = np.sin(2 * np.pi * np.arange(513) / 512)
    assert np.abs(lib_t - true).max() <= 0.5e-8 + 6e-8          # 8 decimals, then float rounding
    xs = np.linspace(-7.0, 14.0, 4001).astype(np.float32)
    s = np.array([orc.arm_sin_f32(x) for x in xs])
    c = np.array([orc.arm_cos_f32(x) for x in xs])
    bound = (2 * np.pi / 512) ** 2 / 8 + 3e-6                   # chord error of a 512-point table
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() <= bound
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() <= bound
    for i in (0, 1, 100, 128, 300, 511):
        x = np.float32(i / 512.0) / np.float32(0.159154943092)
        assert abs(orc.arm_sin_f32(x) - float(lib_t[i])) <= 2e-6
    assert orc.arm_sin_f32(0.0) == 0.0 and orc.arm_cos_f32(0.0) == 1.0
