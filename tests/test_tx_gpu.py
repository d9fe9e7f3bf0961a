"""Parity of the HIP TX (DUC) path with the CPU oracle, through the C ABI: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.lib()
    return ua3reo_ddc


def tone_iq(n, f, amp, fs=48000.0):
    k = np.arange(n)
    return np.stack([np.round(amp * np.cos(2 * np.pi * f * k / fs)),
                     np.round(amp * np.sin(2 * np.pi * f * k / fs))], axis=1).astype(np.int16)


def test_tx_tone_and_random_channels(ddc, orc):
    rng = np.random.default_rng(31)
    n = 40
    iq = np.stack([tone_iq(n, 1000.0, 20000), tone_iq(n, -2500.0, 32767),
                   rng.integers(-32768, 32768, (n, 2)).astype(np.int16),
                   np.full((n, 2), -32768, np.int16)])
    phis = [605867, 85333, 0x3FFFFF, 1]
    d = ddc.Ddc(4)
    d.tune_many(phis)
    dac = d.tx_process(iq)
    for c in range(4):
        want = orc.TxChain(phis[c]).run(iq[c])
        assert np.array_equal(dac[c], want), "channel %d" % c
    assert dac.max() <= 0x3FFF and dac[0].max() > 10000 and dac[0].min() < 6000


@pytest.mark.parametrize("sizes", [[1] * 5, [3, 1, 7], [30, 2]])
def test_tx_streaming_calls_equal_one_call(ddc, orc, sizes):
    rng = np.random.default_rng(32)
    total = sum(sizes)
    iq = rng.integers(-30000, 30000, (2, total, 2)).astype(np.int16)
    phis = [700001, 31]
    d = ddc.Ddc(2)
    d.tune_many(phis)
    out, pos = [], 0
    for s in sizes:
        out.append(d.tx_process(iq[:, pos:pos + s]))
        pos += s
    dac = np.concatenate(out, axis=1)
    for c in range(2):
        assert np.array_equal(dac[c], orc.TxChain(phis[c]).run(iq[c]))


def test_tx_reset_and_phase(ddc, orc):
    iq = tone_iq(6, 700.0, 15000)[None]
    d = ddc.Ddc(1)
    d.tune(0, 123457)
    a = d.tx_process(iq)
    b = d.tx_process(iq)
    d.tx_reset(0)
    c = d.tx_process(iq)
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    d.tx_reset(-1)
    d.tx_set_phase(0, 0x155555)
    o = orc.TxChain(123457)
    o.set_phase(0x155555)
    assert np.array_equal(d.tx_process(iq)[0], o.run(iq[0]))


def test_tx_device_pointer_path_256_channels(ddc, orc):
    torch = pytest.importorskip("torch")
    nch, n = 256, 16
    g = torch.Generator(device="cuda").manual_seed(7)
    iq = torch.randint(-32768, 32768, (nch, n, 2), dtype=torch.int16, device="cuda", generator=g)
    dac = torch.zeros((nch, 1024 * n), dtype=torch.int16, device="cuda")
    phis = [(1234567 * (c + 1)) & 0x3FFFFF for c in range(nch)]
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    torch.cuda.synchronize()
    d.tx_process_dev(iq.data_ptr(), n, dac.data_ptr())
    d.sync()
    host_iq = iq.cpu().numpy()
    host = dac.cpu().numpy().view(np.uint16)
    for c in (0, 7, 128, 255):
        assert np.array_equal(host[c], orc.TxChain(phis[c]).run(host_iq[c]))


def test_tx_many_channels_single_pairs(ddc, orc):
    """Workgroups that straddle many channels: 300 channels, one I/Q pair per call."""
    n_ch = 300
    rng = np.random.default_rng(72)
    iq = rng.integers(-32768, 32768, (n_ch, 4, 2)).astype(np.int16)
    phis = rng.integers(0, 1 << 22, n_ch).astype(np.uint32)
    d = ddc.Ddc(n_ch)
    d.tune_many(phis)
    dac = np.concatenate([d.tx_process(iq[:, k:k + 1]) for k in range(4)], axis=1)
    for c in (0, 1, 127, 128, 255, 256, 299):
        assert np.array_equal(dac[c], orc.TxChain(int(phis[c])).run(iq[c])), c


def test_tx_full_size_properties(ddc, orc):
    """BASELINE config 4 shape (256 channels x 4096 I/Q pairs -> 2^22 DAC words each):
    one call equals two half calls, channels fed the same pairs and tuning word
    agree, every word is a 14-bit value, and three channels match the oracle on
    their first 8 pairs."""
    import torch
    n_ch, n = 256, 4096
    g = torch.Generator(device="cuda").manual_seed(7)
    iq = torch.randint(-20000, 20000, (n_ch, n, 2), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
    iq[255] = iq[0]
    phis = [(7919 * c + 85333) % (1 << 22) for c in range(n_ch)]
    phis[255] = phis[0]

    def run(parts):
        d = ddc.Ddc(n_ch)
        d.tune_many(phis)
        out = torch.zeros((n_ch, 1024 * n), dtype=torch.int16, device="cuda")
        step = n // parts
        for k in range(parts):
            part = iq[:, k * step:(k + 1) * step].contiguous()
            o_k = torch.zeros((n_ch, 1024 * step), dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()  # the ctx has its own stream: torch's fills must have landed
            d.tx_process_dev(part.data_ptr(), step, o_k.data_ptr())
            d.sync()
            out[:, 1024 * k * step:1024 * (k + 1) * step] = o_k
        d.close()
        return out

    a, b = run(1), run(2)
    assert torch.equal(a, b)
    assert torch.equal(a[0], a[255]) and not torch.equal(a[0], a[1])
    assert int(a.min()) >= 0 and int(a.max()) <= 0x3FFF
    for c in (0, 100, 254):
        want = orc.TxChain(phis[c]).run(iq[c, :8].cpu().numpy())
        assert np.array_equal(a[c, :8192].cpu().numpy().view(np.uint16), want)


def test_tx_compensator_output_held_at_the_rails(ddc, orc):
    """Constant inputs of -31232 / -31233 hold both tx_ciccomp phases at -32768 / +32767
    (gain and wrap of the Q14 taps), which drives the tx_cic polynomial to its extremes:
    -2^51 exactly, the smallest start value the integer walk has to represent."""
    n = 40  # the 24-tap phases settle after 24 inputs
    vals = [(-31232, -31232), (-31233, -31233), (-31232, -31233), (31232, 31231)]
    iq = np.stack([np.stack([np.full(n, a), np.full(n, b)], axis=1) for a, b in vals]).astype(np.int16)
    held = orc.TxCicComp().run(iq[0, :, 0].copy())
    assert np.all(held[-8:] == -32768)
    assert np.all(orc.TxCicComp().run(iq[1, :, 0].copy())[-8:] == 32767)
    phis = [605867, 1, 0x200000, 3 << 20]
    d = ddc.Ddc(len(vals))
    d.tune_many(phis)
    dac = d.tx_process(iq)
    d.close()
    for c in range(len(vals)):
        assert np.array_equal(dac[c], orc.TxChain(phis[c]).run(iq[c])), "channel %d" % c
