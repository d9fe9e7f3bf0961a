"""ddc_rx_process_dev_static (the RX entry point that allocates nothing) against
ddc_rx_process_dev: same device inputs, two calls in a row on each of two ctxs
with the same tuning, the second call smaller than the first; outputs must be
identical bit for bit (integer path).  Also the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.lib()  # raises if the HIP library was not built
    return ua3reo_ddc


def test_static_work_area_matches_growing_buffers(ddc, orc):
    torch = pytest.importorskip("torch")
    L = ddc.lib()
    assert L.ddc_rx_static_scratch_bytes() == 32 << 20
    n_ch = 96
    sizes = [1024 * 160, 1024 * 24]          # second call smaller than the first
    assert L.ddc_rx_static_scratch_need(n_ch, sizes[0]) <= L.ddc_rx_static_scratch_bytes()
    rng = np.random.default_rng(2024)
    phis = rng.integers(0, 1 << 22, n_ch).astype(np.uint32)
    phis[0] = 3 << 20                        # sin12 = -2048 throughout: the wrapping product
    ref = ddc.Ddc(n_ch)
    stat = ddc.Ddc(n_ch)
    ref.tune_many(phis)
    stat.tune_many(phis)
    host_adc = []
    for k, n in enumerate(sizes):
        x = rng.integers(-2048, 2048, size=(n_ch, n)).astype(np.int16)
        x[0, :300] = -2048
        host_adc.append(x)
        rb = n // 2 * 3
        rows = np.stack([ddc.pack_adc12(x[c]) for c in range(n_ch)])
        adc = torch.from_numpy(rows).cuda()
        nf = n // 1024
        out_ref = torch.zeros((2, n_ch, nf, 2), dtype=torch.int16, device="cuda")
        out_st = torch.full((2, n_ch, nf, 2), 0x5A5A, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ref.rx_process_dev(adc.data_ptr(), rb, n, out_ref[0].data_ptr(), out_ref[1].data_ptr())
        stat.rx_process_dev_static(adc.data_ptr(), rb, n, out_st[0].data_ptr(), out_st[1].data_ptr())
        ref.sync()
        stat.sync()
        assert torch.equal(out_ref, out_st), "call %d: static work area differs from the growing buffers" % k
        if k == 0:
            first = out_st.cpu().numpy()
    # and both are the reference chain: channel 0 and one more against the oracle over both calls
    second = out_st.cpu().numpy()
    for c in (0, n_ch - 1):
        ws, wv = orc.RxChain(int(phis[c])).run(np.concatenate([host_adc[0][c], host_adc[1][c]]))
        assert np.array_equal(np.concatenate([first[0, c], second[0, c]]), ws)
        assert np.array_equal(np.concatenate([first[1, c], second[1, c]]), wv)
    ref.close()
    stat.close()


def test_static_work_area_rejects_oversized_calls(ddc):
    torch = pytest.importorskip("torch")
    L = ddc.lib()
    # 1024 channels x 2^20 samples (the bench shape) runs as two kernels and fits (about 15 MB), and
    # so does the same plus 1024 samples (two kernels and a small remainder as six); four times
    # the samples do not, nor do 4096 channels x 254 blocks, which go the six-kernel way whole
    # (80 bytes of window moments per block)
    n_ch, n = 1024, 1 << 20
    assert L.ddc_rx_static_scratch_need(n_ch, n) <= L.ddc_rx_static_scratch_need(n_ch, n + 1024) <= L.ddc_rx_static_scratch_bytes()
    assert L.ddc_rx_static_scratch_need(n_ch, 4 * n) > L.ddc_rx_static_scratch_bytes()
    assert L.ddc_rx_static_scratch_need(4 * n_ch, 254 * 512) > L.ddc_rx_static_scratch_bytes()
    d = ddc.Ddc(n_ch)
    adc = torch.zeros((16,), dtype=torch.uint8, device="cuda")   # never read: the call must refuse first
    out = torch.zeros((16,), dtype=torch.int16, device="cuda")
    for samples in (4 * n, 8 * n + 1024):
        with pytest.raises(ddc.DdcError, match="work area"):
            d.rx_process_dev_static(adc.data_ptr(), samples // 2 * 3, samples, out.data_ptr(), out.data_ptr())
    d.close()


def test_static_work_area_runs_the_bench_shape(ddc, orc):
    """1024 channels x 2^20 samples through the entry point that allocates nothing, twice, against
    ddc_rx_process_dev on a second ctx; then a smaller call of each."""
    torch = pytest.importorskip("torch")
    n_ch, n = 1024, 1 << 20
    rb = n // 2 * 3
    phis = [(977 * c + 605867) % (1 << 22) for c in range(n_ch)]
    ref, stat = ddc.Ddc(n_ch), ddc.Ddc(n_ch)
    ref.tune_many(phis)
    stat.tune_many(phis)
    g = torch.Generator(device="cuda").manual_seed(31)
    for samples in (n, n, 1 << 17):
        adc = torch.randint(0, 256, (n_ch, samples // 2 * 3), dtype=torch.uint8, device="cuda", generator=g)
        nf = samples // 1024
        out_ref = torch.zeros((2, n_ch, nf, 2), dtype=torch.int16, device="cuda")
        out_st = torch.full((2, n_ch, nf, 2), 0x5A5A, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ref.rx_process_dev(adc.data_ptr(), samples // 2 * 3, samples, out_ref[0].data_ptr(), out_ref[1].data_ptr())
        stat.rx_process_dev_static(adc.data_ptr(), samples // 2 * 3, samples, out_st[0].data_ptr(), out_st[1].data_ptr())
        ref.sync()
        stat.sync()
        assert torch.equal(out_ref, out_st)
    ref.close()
    stat.close()
